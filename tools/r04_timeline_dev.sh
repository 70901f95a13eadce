# timelines of device-resident calls (kernel trace only): G1 2^20, G2 2^16, G1 2^16, host-ABI G1 2^20
export GPU_MAX_HW_QUEUES=16
R=$PWD; O=$R/gpurun_out/tl; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 4 --warmup 2 --no-cpu-baseline --no-host-abi --no-secondary --sustained 0"
rocprofv3 --kernel-trace --output-format csv -d $O/g1_20 -- python3 $R/bench.py --workload g1msm --log2n 20 $COMMON > $O/g1_20.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/g2_16 -- python3 $R/bench.py --workload g2msm --log2n 16 $COMMON > $O/g2_16.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/g1_16 -- python3 $R/bench.py --workload g1msm --log2n 16 $COMMON > $O/g1_16.log 2>&1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/host -- python3 $R/tools/dbg_host_abi.py 20 g1 8 > $O/host.log 2>&1
cd $R
python3 tools/trace_call.py $O/g1_20 k_msm_reduce_rc 3.6 > $O/timeline_g1_20.txt 2>&1
python3 tools/trace_call.py $O/g2_16 k_msm_window_sum8c_l 1.6 > $O/timeline_g2_16.txt 2>&1
python3 tools/trace_call.py $O/g1_16 k_msm_reduce_rc_p 0.9 > $O/timeline_g1_16.txt 2>&1
python3 tools/trace_call.py $O/host k_msm_reduce_rc 5.2 > $O/timeline_host.txt 2>&1
rm -rf $O/g1_20 $O/g2_16 $O/g1_16 $O/host
tail -30 $O/timeline_g1_20.txt
