set -e
cd /root/repo
timeout -k 10 600 python -u -m pytest tests/test_kernels_gpu.py -x -q --timeout 120 --timeout-method thread -k "layernorm or panel or geglu or split_k or wide" 2>&1 | tail -15
timeout -k 10 600 python -u -m pytest tests/test_model_gpu.py -x -q --timeout 300 --timeout-method thread -k "golden or unet" 2>&1 | tail -15
for f in 1 0; do SDMI_FOLD_LN=$f timeout -k 10 300 python -u tools/step_time.py 8 10 2>&1 | tail -2; done
for f in 1 0; do SDMI_FOLD_LN=$f timeout -k 10 300 python -u tools/step_time.py 1 30 2>&1 | tail -2; done
