"""Drop-in import paths of the reference (`model.conformer`, `model.modules.*`, `model.utils.*`): thin
re-exports of conformer_amd.model, so train.py / test.py / infer.py of the reference import unchanged."""
