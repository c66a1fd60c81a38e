timeout 600 python -m pytest tests/test_dgcnn_gpu.py -x -q -m gpu 2>&1 | tail -2
bash tools/exp_agg_variants.sh 2>&1 | grep -v "sums\|fwd"
