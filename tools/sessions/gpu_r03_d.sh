#!/bin/bash
# GPU session D of round 3: the resident ppg2mel loop (tests, A/B against the chain, wall-clock marks of its roles)
set -u
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ppg2mel_gpu.py tests/test_env_switches_gpu.py -m gpu -q -x --timeout=300 -k "ppg2mel or resident or decode or golden or device_rng" > gpurun_out/pytest_ppg.log 2>&1; echo "pytest_ppg rc=$?"
tail -25 gpurun_out/pytest_ppg.log
MBHIP_DIAG=pr_trace=/tmp/pr_trace.bin timeout 300 python tools/ppg_resident_ab.py > gpurun_out/ppg_resident_ab.log 2>&1; echo "ab rc=$?"
tail -c 4000 gpurun_out/ppg_resident_ab.log
for v in 1 2 3; do echo "variant $v"; MBHIP_DIAG=pr_variant=$v timeout 300 python tools/ppg_resident_ab.py 2>&1 | grep T_enc; done
