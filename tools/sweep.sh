python -m pytest tests/test_hip_parity.py -m gpu -x -q -k rows 2>&1 | tail -4
for w in 4 3; do echo "=== ROWS_MIN_WAVES=$w"; INVPREF_HIPCC_EXTRA="-DROWS_MIN_WAVES=$w" python -c "from invpref_kdd_2022_amd import build; build.build(force=True)"; python tools/kbench.py 2>&1 | grep "rows per" ; done
python tools/stamps.py 2>&1 | grep -v amdgpu | head -24
