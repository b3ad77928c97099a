python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head
python tools/kbench.py 2>&1 | grep -E "rows per"
python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200
