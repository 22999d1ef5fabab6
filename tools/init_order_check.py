import sys
sys.path.insert(0, ".")
mode = sys.argv[1]
import torch
if mode == "torch_first":
    print("torch first:", torch.cuda.is_available())
from vittracker_amd import native
m = native.Model(64, 128, max_batch=2)
print("after Model:", torch.cuda.is_available(), torch.cuda.device_count())
x = torch.zeros(4, device="cuda")
print("tensor ok", x.device)
