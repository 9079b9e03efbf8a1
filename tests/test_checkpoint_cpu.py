"""Reference checkpoint layout (checkpoint.py:7-35, manager.py:23-47) round-trips through the module mirror on CPU."""
import torch

from conformer_amd import checkpoint as C
from conformer_amd.optim import FusedAdam
from model.conformer import Conformer


def _model():
    torch.manual_seed(0)
    return Conformer(37, 80, 1, 32, 4, 7, 16, 1, 0.0)


def test_reference_layout_round_trip(tmp_path):
    m = _model()
    opt = torch.optim.Adam(m.parameters(), lr=2e-5)            # what train.py:188 builds
    sched = torch.optim.lr_scheduler.StepLR(opt, 10)
    path = str(tmp_path / "1000.pt")
    C.save_checkpoint(path, m, opt, sched, n_steps=1000, n_epochs=3)
    m2 = _model()
    with torch.no_grad():
        for p in m2.parameters():
            p.add_(1.0)
    opt2 = FusedAdam(m2.parameters(), lr=1e-3)                 # the fused optimiser accepts the stock optimiser's state
    steps, epochs = C.load_checkpoint(path, m2, opt2, torch.optim.lr_scheduler.StepLR(opt2, 10))
    assert (steps, epochs) == (1000, 3)
    for (k, a), (_, b) in zip(m.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    assert opt2.param_groups[0]["lr"] == 2e-5


def test_ddp_prefix_conversion():
    m = _model()
    sd = m.state_dict()
    ddp_sd = C.convert_prefix(sd, ddp=True)
    assert C.is_ddp_state_dict(ddp_sd) and not C.is_ddp_state_dict(sd)
    assert list(C.convert_prefix(ddp_sd, ddp=False)) == list(sd)
    m2 = _model()
    C.load_model(ddp_sd, m2, world_size=1)                     # a multi-GPU checkpoint into a single-GPU model
    assert all(torch.equal(a, b) for a, b in zip(m.state_dict().values(), m2.state_dict().values()))
