python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|^E " | head
echo "--- Yahoo"; python tools/kbench.py 2>&1 | grep -E "graph:"
echo "--- MovieLens-scale"; python tools/kbench.py 6040 3706 8 128 65536 2>&1 | grep -E "graph:" 
echo "--- MIND-scale"; python tools/kbench.py 50000 51283 16 256 262144 2>&1 | grep -E "graph:"
