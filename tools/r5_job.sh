mkdir -p gpurun_out
for rep in 1 2; do
bash tools/exp_env.sh "-" "DSL_TUNE=tower_slots=64" "DSL_TUNE=tower_slots=80" "DSL_TUNE=tower_slots=96" "DSL_TUNE=tail_slots=160" "DSL_TUNE=tail_slots=224" "DSL_TUNE=tail_slots=256" "DSL_TUNE=lib.wgrad_slots=112" "DSL_TUNE=lib.wgrad_slots=144" "DSL_TUNE=img_split_bwd=234" "DSL_TUNE=img_split_bwd=3" "DSL_TUNE=img_split=34"
done 2>&1 | tee gpurun_out/r05_knob_sweep.txt
