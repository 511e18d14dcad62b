#!/bin/bash
# The two roofline peaks as measured on the box ("verify the two peaks ... and report them"):
#   fp32 MFMA: tools/probes/mfma_probe.hip V0 = a register-only v_mfma_f32_16x16x4_f32 stream on every SIMD
#   HBM: a 2 GiB device-to-device copy (read + write bytes) and a 2 GiB fill (write only)
# usage (through gpurun, from the repo root): bash tools/peaks.sh > gpurun_out/peaks.txt
set -e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_probe tools/probes/mfma_probe.hip 2>/dev/null
echo "# fp32 MFMA stream, zero operands (clocks higher, MI355X_MICROARCH.md DVFS note)"
/tmp/mfma_probe | head -2
echo "# fp32 MFMA stream, random operands"
PROBE_RANDOM=1 /tmp/mfma_probe | head -2
python - <<'PY'
import torch, time
a = torch.empty(1 << 29, device="cuda", dtype=torch.float32)   # 2 GiB
b = torch.empty_like(a)
a.fill_(1.0)
for name, fn, nbytes in (("copy (read+write)", lambda: b.copy_(a), 2 * a.numel() * 4), ("fill (write)", lambda: b.fill_(2.0), a.numel() * 4)):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    e1.synchronize()
    ms = e0.elapsed_time(e1) / 10
    print("# HBM %s of 2 GiB: %.3f ms  %.0f GB/s" % (name, ms, nbytes / ms / 1e6))
PY
