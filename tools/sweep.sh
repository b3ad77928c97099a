python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head
python tools/kbench.py 2>&1 | grep -E "rows per"
for sr in 32 128; do echo "stream rows $sr"; INVPREF_PLAN_STREAM_ROWS=$sr python tools/kbench.py 2>&1 | grep -E "rows per" | head -2; done
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
