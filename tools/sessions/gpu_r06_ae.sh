#!/bin/bash
# round 6: knock-out builds of conv_split_tm at the narrow upsampler shapes
S="64,64,20000,3 128,256,5000,3 256,640,1000,3 512,1024,400,1,split 1024,512,400,3,split"
python tools/ctm_bench.py 20 $S
for b in 1 2 3 4 8 12; do MBHIP_LIB=build_variants/libmbhip_ctd$b.so python tools/ctm_bench.py 20 $S; done
