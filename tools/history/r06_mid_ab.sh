set -u
timeout 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "runahead" 2>&1 | tail -2
for spec in "24 1.0" "32 1.0" "40 1.0" "48 1.0" "12 4.0" "16 4.0"; do timeout 300 python tools/mid_ab.py $spec 300 ";CAR_NO_RUNAHEAD=1;" 2>/dev/null | cut -c1-200; done
