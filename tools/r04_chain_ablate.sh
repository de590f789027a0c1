export TMPDIR=/tmp
python tools/bench_kernels.py chain 2>&1 | grep -v amdgpu.ids
for v in 1 2 4 6 7; do echo "== CHAIN_ABL=$v (1 no W loads, 2 no MFMAs, 4 no stage writes)"; LAMP_HIP_LIBRARY=$PWD/lamp_amd/build/liblamp_abl$v.so python tools/bench_kernels.py chain 2>&1 | grep "chain launch\|fc  \|W1  \|clock"; done
