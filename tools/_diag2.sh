MBHIP_RNN_TS2=1 timeout 300 python -m pytest tests/test_wavernn_gpu.py -m gpu -q -k "batch" 2>&1 | tail -3
bash tools/_diag.sh "$@"
