python -m pytest tests/test_gpu_rvq.py -q 2>&1 | tail -4
for i in 1 2; do
HILC_RVQ_VALU=1 python tools/layer_profile.py 2>&1 | grep -E "rvq|total"
python tools/layer_profile.py 2>&1 | grep -E "rvq|total"
HILC_RVQ_VALU=1 python tools/layer_profile.py --model hil_music 2>&1 | grep -E "rvq|total"
python tools/layer_profile.py --model hil_music 2>&1 | grep -E "rvq|total"
done
