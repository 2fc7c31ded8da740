for n in 6144 8192 10000 11264; do
 for cfg in "GPP_BORDER_RL=0 GPP_BORDER_FULL=0" "GPP_BORDER_RL=1 GPP_BORDER_FULL=1" "GPP_BORDER_RL=0 GPP_BORDER_FULL=1" "GPP_BORDER_RL=0 GPP_BORDER_FULL=0"; do echo -n "N=$n $cfg: "; env $cfg timeout 200 python tools/bench_stages.py $n 8 4 2>&1 | grep "potrf\|total" | tr '\n' ' '; echo; done
done
