#!/bin/bash
# round 6 session C: the fp32 generators on the time-major split pairs (parity suites + forward time against the channel-major plan)
mkdir -p gpurun_out
python -m pytest tests/test_gan_gpu.py tests/test_vits_gpu.py tests/test_resblock_pair_split_gpu.py tests/test_pipeline_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r06_c_pytest.log
cat gpurun_out/r06_c_pytest.log
for f in all cm; do
  echo "== MBHIP_GAN_FUSE=$f" >> gpurun_out/r06_c_gan.log
  MBHIP_GAN_FUSE=$f python tools/gan_run.py hifigan f32 32 200 5 >> gpurun_out/r06_c_gan.log 2>&1
  MBHIP_GAN_FUSE=$f python tools/gan_run.py fregan f32 8 1000 3 >> gpurun_out/r06_c_gan.log 2>&1
done
cat gpurun_out/r06_c_gan.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d /tmp/prof_c -o gan -- python $GRAFT_REPO_ROOT/tools/gan_run.py hifigan f32 32 200 5 > /dev/null 2>&1
f=$(find /tmp/prof_c -name "*kernel_stats.csv" | head -1); cp "$f" $GRAFT_REPO_ROOT/gpurun_out/r06_c_hifigan_f32_kernel_stats.csv
head -30 $GRAFT_REPO_ROOT/gpurun_out/r06_c_hifigan_f32_kernel_stats.csv | cut -c1-200
