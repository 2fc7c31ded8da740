for n in 4096 6144 8192 10000 11264; do
 for rl in 1 0; do echo -n "N=$n RL=$rl: "; GPP_BORDER_RL=$rl timeout 200 python tools/bench_stages.py $n 8 4 2>&1 | grep "potrf\|trtri\|total" | tr '\n' ' '; echo; done
done
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_kernels_r3.py -m gpu -x -q 2>&1 | tail -2
