python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "g5" 2>&1 | grep -E "passed|failed|^E " | head
echo "--- MovieLens-scale (U=6040 I=3706 E=8 D=128 B=65536)"; python tools/kbench.py 6040 3706 8 128 65536 2>&1 | grep -E "graph:|rows per" | head -4
echo "--- MIND-scale (U=50000 I=51283 E=16 D=256 B=262144)"; python tools/kbench.py 50000 51283 16 256 262144 2>&1 | grep -E "graph:|rows per" | head -4
