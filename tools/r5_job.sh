mkdir -p gpurun_out
for rep in 1 2 3; do
bash tools/exp_env.sh "-" "DSL_HIP_LIB=$PWD/dsl_amd/lib_ta128/libdsl_hip.so"
done 2>&1 | tee gpurun_out/r05_gn_ta128_ab.txt
