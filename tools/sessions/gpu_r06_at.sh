#!/bin/bash
# round 6: resident-window conv kernel: tests, microbench and whole-path A/B against MBHIP_DIAG=ctm_nores
python -m pytest tests/test_conv_split_tm_gpu.py -x -q -m gpu 2>&1 | tail -3
S="256,640,1000,3 512,1280,200,3 512,1024,400,1,split 80,1024,400,9 80,512,200,7"
for d in "" ctm_nores; do echo "== MBHIP_DIAG=$d"; MBHIP_DIAG=$d python tools/ctm_bench.py 20 $S 2>&1 | cut -c1-100; done
for d in "" ctm_nores "" ctm_nores; do
  echo "== MBHIP_DIAG=$d"
  MBHIP_DIAG=$d python tools/gan_run.py hifigan f32 32 200 20 2>&1 | tail -1
  MBHIP_DIAG=$d python tools/taco_gen_time.py 2>&1 | tail -2
done
python -m pytest tests/test_gan_gpu.py tests/test_gan_rb2_gpu.py tests/test_vits_gpu.py tests/test_pipeline_gpu.py tests/test_tacotron_gpu.py -x -q -m gpu 2>&1 | tail -3
