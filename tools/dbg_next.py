import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from safetensors.torch import load_file
from rvdd_release_amd.runtime import RvddRuntime
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sd = load_file(os.path.join(R, "weights", "recurrent-ConvNeXtUnet-iso3200.safetensors"))
torch.manual_seed(0)
x = torch.randn(1, 6, 32, 32, device="cuda") * 0.3
outs = {}
for split in (0, 1):
    rt = RvddRuntime("next", 0, 1, 32, 32, 0)
    rt.set_option("next_split", split)
    rt.set_option("next_pool", 0)
    rt.load_state_dict(sd)
    o = rt.unet_forward(x, None)
    o = o[0] if isinstance(o, tuple) else o
    outs[split] = o.clone()
    print("split", split, "nan count", int(torch.isnan(o).sum()), "of", o.numel(), "absmax", float(o[~torch.isnan(o)].abs().max()) if (~torch.isnan(o)).any() else None)
    rt.close()
d = (outs[0] - outs[1]).abs()
print("max diff", float(d[~torch.isnan(d)].max()) if (~torch.isnan(d)).any() else None)
