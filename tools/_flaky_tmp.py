import sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_next_rows_gpu import test_training_trajectory_fused_adam_equals_torch_adam as t
dev = torch.device("cuda:0")
bad = 0
for i in range(12):
    for amp in (None, torch.bfloat16):
        try:
            t(dev, amp)
        except AssertionError as e:
            bad += 1
            print("FAIL", i, amp, str(e)[:200])
print("failures:", bad)
