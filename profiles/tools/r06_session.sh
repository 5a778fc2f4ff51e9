#!/bin/bash
# round 6, session A: VALU calibration + stall breakdown of the product kernels
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r6a
mkdir -p $O
cd $R
timeout 300 ./profiles/microbench/valu_calibration.bin > $O/valu_calibration.txt 2>&1
tail -40 $O/valu_calibration.txt
timeout 300 ./profiles/microbench/lane_split_gate.bin > $O/lane_split_gate.txt 2>&1
cat $O/lane_split_gate.txt
cd /tmp && export TMPDIR=/tmp
(rocprofv3-avail list 2>&1 || rocprofv3 -L 2>&1) > $O/counters_avail.txt
grep -c . $O/counters_avail.txt
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_INST_CYCLES_VMEM -d /tmp/p_cal -o r -- $R/profiles/microbench/valu_calibration.bin pmc > $O/cal_pmc_stdout.txt 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_cal2 -o r -- $R/profiles/microbench/valu_calibration.bin pmc > $O/cal_pmc2_stdout.txt 2>&1
cd $R
python - <<PY > $O/cal_pmc_summary.txt 2>&1
import sys, glob
sys.path.insert(0, "profiles/tools")
import pmc_summary as ps
for d in ("/tmp/p_cal", "/tmp/p_cal2"):
    for db in glob.glob(d + "/**/*_results.db", recursive=True):
        r = ps.read_db(db)
        for k in sorted(r, key=lambda s: int(s[2:-1]) if s.startswith("k<") else 999):
            print(d, k, {c: round(v, 1) for c, v in r[k].items()})
PY
cat $O/cal_pmc_summary.txt | head -80
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $O/bench.json 2> $O/bench.err
cat $O/bench.json | head -c 1500
cd /tmp
P="--steps 4 --warmup 1 --repeats 1 --no-cpu-baseline --no-roofline --no-secondary"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d /tmp/p_s1 -o r -- python $R/bench.py $P > /dev/null 2> $O/s1.err
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU -d /tmp/p_s2 -o r -- python $R/bench.py $P > /dev/null 2> $O/s2.err
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT -d /tmp/p_s3 -o r -- python $R/bench.py $P > /dev/null 2> $O/s3.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d /tmp/p_s4 -o r -- python $R/bench.py $P > /dev/null 2> $O/s4.err
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_SALU SQ_WAVES -d /tmp/p_s5 -o r -- python $R/bench.py $P > /dev/null 2> $O/s5.err
cd $R
python - <<PY > $O/stall_summary.txt 2>&1
import sys, glob, json
sys.path.insert(0, "profiles/tools")
import pmc_summary as ps
out = {}
for d in ("/tmp/p_s1", "/tmp/p_s2", "/tmp/p_s3", "/tmp/p_s4", "/tmp/p_s5"):
    for db in glob.glob(d + "/**/*_results.db", recursive=True):
        for k, v in ps.read_db(db).items():
            out.setdefault(k, {}).update(v)
json.dump(out, open("$O/stall_pmc.json", "w"), indent=1, sort_keys=True)
for k in sorted(out, key=lambda k: -out[k].get("SQ_WAVE_CYCLES", 0) * out[k]["calls"])[:9]:
    print(k, json.dumps({c: round(x, 1) for c, x in out[k].items()}))
PY
cat $O/stall_summary.txt
# the file bench.py's roofline.valu reads: the table of step 1 x the instruction mix just collected (copy it to profiles/ to commit it)
python profiles/tools/valu_calibration.py $O/valu_calibration.txt $O/stall_pmc.json --workload config3_cube128 --dtype f32 --out $O/r06_valu_calibration.json > $O/valu_calibration_summary.txt 2>&1
cat $O/valu_calibration_summary.txt
tail -3 $O/s1.err $O/s2.err $O/s3.err $O/s4.err $O/s5.err
# ---- kernel A/B of the prepared variants on bit-identical inputs, then the headline, then parity of the variants
cd $R
timeout 600 python profiles/tools/replay_ab.py --reps 30 default bufio g3 g3buf g3bufpk2 pk1 pk1buf pk3buf > $O/replay_ab.txt 2>&1
tail -14 $O/replay_ab.txt
timeout 900 python profiles/tools/ab.py $O/ab --reps 2 --args "--steps 20 --warmup 5 --no-cpu-baseline --no-secondary" default bufio g3buf g3bufpk2 pk1buf > $O/ab.txt 2>&1
tail -8 $O/ab.txt
for v in g3bufpk2 pk1buf; do
  PLMPM_LIB=exp_libs/libplmpm_$v.so timeout 900 python -m pytest tests/test_gpu_substep.py tests/test_gpu_edge_sizes.py tests/test_gpu_rollout.py -q -x 2>&1 | tail -4 > $O/pytest_$v.txt
  cat $O/pytest_$v.txt
done
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -6 > $O/pytest_gpu_default.txt
cat $O/pytest_gpu_default.txt
