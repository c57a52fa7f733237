mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/gputests.log 2>&1 || { tail -30 gpurun_out/gputests.log; exit 1; }
tail -2 gpurun_out/gputests.log
ABSTEPS="--steps 500 --warmup 20" bash tools/ab_variants.sh "scene3_4k scene1_4k tex4_4k scene2_4k spheres10k_4k" build_variants/base.so build_variants/diag.so 2>&1 | tee gpurun_out/ab_diag.txt
ABSTEPS="--steps 2 --warmup 1" bash tools/ab_variants.sh "spheres10k_8k" build_variants/base.so build_variants/diag.so 2>&1 | tee -a gpurun_out/ab_diag.txt
