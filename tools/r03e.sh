#!/bin/bash
# round 3, GPU call e: profiles of the round (kernel stats + PMC), the resize-mode sweep, device clock
O=gpurun_out/r03e; mkdir -p $O
python - <<'PY' | tee $O/device.txt
import torch
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if not k.startswith("_") and k not in ("uuid",)})
PY
python -m pytest tests/test_gpu_dup_heavy.py -m gpu -x -q 2>&1 | tail -2
timeout 900 python tools/sweep_resize_modes.py --modes 0,2,4,5,6 > $O/resize_sweep.txt 2>&1
tail -40 $O/resize_sweep.txt
bash tools/profile_round.sh r03e/prof > /dev/null 2>&1
ls $O/prof
