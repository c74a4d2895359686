# usage: bash tools/history/r06_posdbg.sh <alt dir names...>   (libraries built from tools/history/r06_posdbg_*.diff with -DCAR_POSDBG)
mkdir -p gpurun_out/r06_posdbg
for A in "$@"; do
for B in 384 768; do TWIN_LIB=controlar_amd/csrc/$A/libcontrolar_hip.so CALLS=${CALLS:-4} timeout 300 python tools/history/r06_posdbg_probe.py $B 96 > gpurun_out/r06_posdbg/${A}_b$B.txt 2>&1; done
echo "== $A"; grep -h "mismatches\|call\|consensus\|rror\|first " gpurun_out/r06_posdbg/${A}_b*.txt | grep -v "mismatches 0 of 0" 
done
