export GPU_MAX_HW_QUEUES=16
R=$PWD; O=$R/gpurun_out/tls; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
COMMON="--steps 4 --warmup 2 --no-cpu-baseline --no-host-abi --no-secondary --sustained 0"
rocprofv3 --kernel-trace --output-format csv -d $O/g1_7 -- python3 $R/bench.py --workload g1msm --log2n 7 $COMMON > $O/g1_7.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/g2_7 -- python3 $R/bench.py --workload g2msm --log2n 7 $COMMON > $O/g2_7.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/g1_12 -- python3 $R/bench.py --workload g1msm --log2n 12 $COMMON > $O/g1_12.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $O/p3 -- python3 $R/bench.py --workload pairing --log2n 3 $COMMON > $O/p3.log 2>&1
cd $R
for t in g1_7 g2_7 g1_12; do k=$(grep -o "k_msm_reduce[a-z0-9_]*" $O/$t/*/*kernel_trace.csv | sort | uniq -c | sort -rn | head -1 | awk '{print $2}' | sed 's/.*://'); python3 tools/trace_call.py $O/$t ${k:-k_msm_reduce} 0.7 > $O/timeline_$t.txt 2>&1; done
python3 tools/trace_call.py $O/p3 k_pair_tree2 0.9 > $O/timeline_p3.txt 2>&1
rm -rf $O/g1_7 $O/g2_7 $O/g1_12 $O/p3
tail -28 $O/timeline_g1_7.txt
