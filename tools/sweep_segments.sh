set -e
cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for seg in 4 8 16 32 64 128 256; do
  echo "== seg $seg"; python bench.py --steps 10 --warmup 2 --workload c3 --segment-rows $seg --no-cpu-baseline --no-extra 2>&1 | tail -1
done
for seg in 8 32 256; do
  echo "== c2 seg $seg"; python bench.py --steps 10 --warmup 2 --workload c2 --segment-rows $seg --no-cpu-baseline --no-extra 2>&1 | tail -1
done
echo "== target default"; python bench.py --steps 10 --warmup 2 2>&1 | tail -1
