#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root:  bash tools/profile_final.sh <tag>
# Everything profiles/ records for a round: kernel trace + FETCH_SIZE / WRITE_SIZE passes + one SQ-counter pass of the default
# bench command, the counter calibration, and un-profiled bench lines of all BASELINE configurations, size sweep, frame probe.
set -u
TAG=$1
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd $ROOT
bash tools/profile_gpu.sh ${TAG} --steps 200 --warmup 10
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  -d $OUT/${TAG}_sq -o s -- python $ROOT/bench.py --no-cpu-baseline --no-drop-in --steps 5 --warmup 2 > /dev/null 2> $OUT/${TAG}_sq.err
cd $ROOT
python tools/rocpd_summary.py $(find $OUT/${TAG}_sq -name '*.db') > $OUT/${TAG}_sq_counters.txt
rm -rf $OUT/${TAG}_sq
bash tools/calib_gpu.sh ${TAG}
python bench.py --steps 500 --warmup 20 > $OUT/${TAG}_bench_default.json 2> $OUT/${TAG}_bench_default.err
python bench.py --preset cfg2 --steps 300 --warmup 20 --no-drop-in > $OUT/${TAG}_bench_ionic_cfg2.json 2>> $OUT/${TAG}_bench_default.err
python bench.py --preset cfg4 --steps 200 --warmup 20 --no-drop-in --no-cpu-baseline > $OUT/${TAG}_bench_cfg4_8frames_1gpu.json 2>> $OUT/${TAG}_bench_default.err
python bench.py --preset cfg4 --frame-batch streams --steps 200 --warmup 20 --no-drop-in --no-cpu-baseline > $OUT/${TAG}_bench_cfg4_streams.json 2>> $OUT/${TAG}_bench_default.err
python bench.py --preset cfg5 --steps 100 --warmup 10 --no-drop-in > $OUT/${TAG}_bench_dispersion_cfg5.json 2>> $OUT/${TAG}_bench_default.err
python bench.py --launch eager --steps 200 --warmup 20 --no-drop-in --no-cpu-baseline > $OUT/${TAG}_bench_eager.json 2>> $OUT/${TAG}_bench_default.err
python tools/size_sweep.py > $OUT/${TAG}_size_sweep.txt 2>&1
python tools/frames_probe.py > $OUT/${TAG}_frames_probe.txt 2>&1
python tools/time_stress.py > $OUT/${TAG}_stress.txt 2>&1
ls -la $OUT | grep ${TAG}
# round 3: device neighbour stream + live-bin step
python bench.py --neighbors stream --steps 500 --warmup 20 --no-drop-in --no-cpu-baseline > $OUT/${TAG}_bench_stream.json 2>> $OUT/${TAG}_bench_default.err
python tools/time_refresh.py > $OUT/${TAG}_refresh.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_live -o l -- python $ROOT/tools/prof_live.py > /dev/null 2>&1)
python tools/rocpd_summary.py $(find $OUT/${TAG}_live -name '*.db') > $OUT/${TAG}_kernel_stats_live_step.txt
rm -rf $OUT/${TAG}_live
ls -la $OUT | grep ${TAG}
# round 3: the reference call sequence through the compiled front end and through the Python nodes
python tools/prof_dropin.py 300 > $OUT/${TAG}_prof_dropin_front.txt 2>&1
MIPME_FRONT=0 python tools/prof_dropin.py 300 > $OUT/${TAG}_prof_dropin_python_nodes.txt 2>&1
python tools/prof_size.py 56 256 > /dev/null 2>&1 || true
ls -la $OUT | grep ${TAG}
# kernel traces of the fp64 configurations and of the reference call sequence
for cfg in cfg2 cfg4; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_$cfg -o t -- python $ROOT/bench.py --preset $cfg --steps 100 --warmup 10 --no-drop-in --no-cpu-baseline > /dev/null 2>&1)
  python tools/rocpd_summary.py $(find $OUT/${TAG}_$cfg -name '*.db') > $OUT/${TAG}_kernel_stats_$cfg.txt
  rm -rf $OUT/${TAG}_$cfg
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_di -o t -- python $ROOT/tools/prof_dropin_kernels.py 300 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find $OUT/${TAG}_di -name '*.db') > $OUT/${TAG}_kernel_stats_dropin.txt
rm -rf $OUT/${TAG}_di
# round 4: the energy step with the whole autograd contract (E, F, dE/dq, dE/dcell), binned and live; the TuningTimings protocol
python tools/check_contract.py > $OUT/${TAG}_check_contract.txt 2>&1
for m in F Fqc; do
  for live in 0 1; do
    (cd /tmp && MODE=$m LIVE=$live rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_c_$m$live -o t -- python $ROOT/tools/prof_contract.py > /dev/null 2>&1)
    python tools/rocpd_summary.py $(find $OUT/${TAG}_c_$m$live -name '*.db') > $OUT/${TAG}_kernel_stats_contract_${m}_live$live.txt
    rm -rf $OUT/${TAG}_c_$m$live
  done
done
PROFILE=1 python tools/prof_protocol.py 200 > $OUT/${TAG}_protocol_prof.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_pp -o t -- python $ROOT/tools/prof_protocol.py 200 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find $OUT/${TAG}_pp -name '*.db') > $OUT/${TAG}_kernel_stats_protocol.txt
rm -rf $OUT/${TAG}_pp
# round 4: compiled front end with charges / cell gradients; a new list tensor every call
python tools/check_front_contract.py > $OUT/${TAG}_check_front_contract.txt 2>&1
python tools/prof_cold.py 100 > $OUT/${TAG}_cold_list.txt 2>&1
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_cold -o t -- python $ROOT/tools/prof_cold.py 100 > /dev/null 2>&1)
python tools/rocpd_summary.py $(find $OUT/${TAG}_cold -name '*.db') > $OUT/${TAG}_kernel_stats_cold_list.txt
rm -rf $OUT/${TAG}_cold
python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_driver_flags.json 2>> $OUT/${TAG}_bench_default.err
ls -la $OUT | grep ${TAG}
# cfg5 counter passes, the plane workgroups' timeline
bash tools/profile_gpu.sh ${TAG}_cfg5 --preset cfg5 --steps 50 --warmup 5
cd /tmp
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY \
  -d $OUT/${TAG}_sq5 -o s -- python $ROOT/bench.py --preset cfg5 --no-cpu-baseline --no-drop-in --steps 5 --warmup 2 > /dev/null 2> $OUT/${TAG}_sq5.err
cd $ROOT
python tools/rocpd_summary.py $(find $OUT/${TAG}_sq5 -name '*.db') > $OUT/${TAG}_sq_counters_cfg5.txt
rm -rf $OUT/${TAG}_sq5
if [ -f torch-pme_amd/libmipme_timeline.so ]; then
  MIPME_LIB=$ROOT/torch-pme_amd/libmipme_timeline.so python tools/r05/plane_timeline.py water > $OUT/${TAG}_plane_timeline.txt 2>&1
fi
ls -la $OUT | grep ${TAG}
# round 6: the distributed path with one rank over RCCL (every exchange protocol), the order / scheme sweep, measured errors against
# the reference's own full-size numbers
bash tools/r06/dist_matrix.sh 1 > /dev/null 2>&1
cp $OUT/r06_dist_1rank.txt $OUT/${TAG}_dist_1rank.txt
python tools/r06/order_sweep.py > $OUT/${TAG}_order_sweep.txt 2>&1
python tools/r06/fullsize_errors.py > $OUT/${TAG}_fullsize_errors.txt 2>&1
ls -la $OUT | grep ${TAG}
