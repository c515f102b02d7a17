set -u
export TMPDIR=/tmp
out=$PWD/gpurun_out/certprobe
mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out/seq -- python tools/profile_target.py B=64 reps=1 noov=1 > $out/seq.log 2>&1
python tools/trace_iterations.py $out/seq nn_ball_lds nn_certify nn_ball_listed accumulate finalize > $out/iter.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/f -- python tools/profile_target.py B=64 reps=1 noov=1 > $out/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/w -- python tools/profile_target.py B=64 reps=1 noov=1 > $out/w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $out/s -- python tools/profile_target.py B=64 reps=1 noov=1 > $out/s.log 2>&1
python tools/pmc_summary.py $out/f nn_certify accumulate "nn_ball" > $out/pmc.txt
python tools/pmc_summary.py $out/w nn_certify accumulate "nn_ball" >> $out/pmc.txt
python tools/pmc_summary.py $out/s nn_certify accumulate "nn_ball" >> $out/pmc.txt
rm -rf $out/seq $out/f $out/w $out/s
cat $out/iter.txt $out/pmc.txt
