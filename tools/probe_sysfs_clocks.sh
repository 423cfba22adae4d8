#!/bin/bash
# What the GPU box exposes of the GPU's clock and power (bench.py samples these around its timed regions)
for c in /sys/class/drm/card*; do
  [ -e $c/device/pp_dpm_sclk ] || continue
  echo "== $c -> $(readlink -f $c/device)"
  cat $c/device/pp_dpm_sclk 2>&1 | head -12
  for h in $c/device/hwmon/hwmon*; do
    echo "-- $h"; ls $h | tr '\n' ' '; echo
    for f in power1_average power1_input power1_cap freq1_input freq1_label freq2_input freq2_label temp1_input; do [ -e $h/$f ] && echo "$f = $(cat $h/$f 2>&1)"; done
  done
  for f in gpu_busy_percent mem_busy_percent current_link_speed pp_dpm_mclk pp_dpm_fclk; do [ -e $c/device/$f ] && { echo "$f:"; cat $c/device/$f 2>&1 | head -6; }; done
done
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if "pci" in k or k in ("name", "clock_rate", "multi_processor_count")})
PY
