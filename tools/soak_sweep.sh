cd $GRAFT_REPO_ROOT
# the random option sweep with more cases and other seeds than the suite's: soak_sweep.sh [seeds...] (default 1 2 3)
for seed in ${@:-1 2 3}; do
DEMOD_FUZZ_CASES=1500 DEMOD_FUZZ_SEED=$seed timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -q -x -k random_option_sets -p no:cacheprovider 2>&1 | tail -6
done
