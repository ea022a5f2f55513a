# Final batch of a round on the final tree -> gpurun_out/<name>/   usage: bash tools/gpu_final.sh <name>  (through gpurun; ~7.5 GPU-minutes)
export TMPDIR=/tmp
R=$PWD; O=$R/gpurun_out/${1:-final}; mkdir -p $O
bash tools/gpu_run.sh ${1:-final} tests
bash tools/gpu_run.sh ${1:-final} traffic:chair traffic:trex traffic:stress
bash tools/gpu_run.sh ${1:-final} "pmc:chair:sq1:SQ_WAVES+SQ_WAVE_CYCLES+SQ_BUSY_CYCLES+SQ_WAIT_ANY+SQ_WAIT_INST_ANY+SQ_ACTIVE_INST_ANY+SQ_INSTS_VALU+SQ_INSTS_SALU" "pmc:chair:tcc:TCC_HIT_sum+TCC_MISS_sum+TCC_REQ_sum" "pmc:chair:lat:VmemLatency" "pmc:chair:sq3:SQ_VALU_MFMA_BUSY_CYCLES+SQ_INSTS_MFMA+SQ_INSTS_VALU_MFMA_MOPS_F16+SQ_BUSY_CU_CYCLES+SQ_THREAD_CYCLES_VALU"
python tools/pmc_instr.py /tmp/pmc_chair_sq1 3 k_ > $O/instr_per_kernel_sq1.txt 2>&1
bash tools/gpu_run.sh ${1:-final} bench:chair bench:chair:--steps+20+--warmup+5 bench:chair:--steps+20+--warmup+5+--no-extras+--no-cpu-baseline bench:chair:--steps+20+--warmup+5+--no-extras+--no-cpu-baseline+--prime+0
bash tools/gpu_run.sh ${1:-final} bench:trex bench:stress bench:chair:--no-extras+--no-cpu-baseline+--lanes+1 bench:chair:--no-extras+--no-cpu-baseline+--lanes+2 bench:chair:--no-extras+--no-cpu-baseline+--no-d2h
bash tools/gpu_run.sh ${1:-final} stats:chair stats:trex stats:stress eager:chair sim clocks
PN_SIM_FORM=csr python tools/time_sim.py 2>&1 | grep -v amdgpu.ids > $O/time_sim_csr.txt
PN_LIB_PATH=$R/pienerf_amd/lib/variants/simstamps.so python tools/sim_stamps.py 2>&1 | grep -v amdgpu.ids > $O/sim_stamps.txt
python bench.py --parallelism tile --steps 40 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_tile_n1.json 2>/dev/null
timeout 900 python tools/soak.py --frames 3000 > $O/soak_3000.json 2> $O/soak.err; tail -c 300 $O/soak_3000.json
ls $O | wc -l
python tools/trex_region_trajectory.py --frames 300 --every 10 > $O/trex_region_trajectory.json 2>/dev/null
