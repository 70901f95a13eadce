#!/bin/bash
R=$PWD
for L in lib_r1outl.so; do
  for rb in 140 200 256 400; do
      EIP2537_REDUCE_BLOCKS=$rb EIP2537_HIP_LIB=$PWD/variants/$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-secondary --no-host-abi 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L rb $rb', 'ms/step %.3f'%d['ms_per_step'], 'min %.3f'%d['step_ms']['min'], 'pipeline %.3f'%d['roofline']['device_pipeline_ms'], 'accum %.3f'%d['roofline']['kernel_ms'], 'exact', d['bit_exact_vs_golden'])"
  done
done
cd /tmp && export TMPDIR=/tmp
for rb in 140 256; do
EIP2537_REDUCE_BLOCKS=$rb rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_IFETCH SQ_WAVES SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/r2h/sq_rb$rb -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary --no-host-abi > $R/gpurun_out/r2h/sq_rb$rb.log 2>&1
done
cd $R
python3 - <<'PY'
import csv,glob,collections
for rb in (140,256):
    d=collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("gpurun_out/r2h/sq_rb%d/*/*counter_collection.csv"%rb):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"].split("(")[0].replace("eip::","")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k,v in d.items():
        m={c:max(x) for c,x in v.items()}
        if "reduce" not in k and "accum" not in k: continue
        wc=m["SQ_WAVE_CYCLES"]
        print("rb%d %-30s waves %6d cyc/wave %9.0f valu_active %4.1f%% wait_any %4.1f%% wait_inst %4.1f%% valu/wave %8.0f salu/wave %7.0f ifetch/wave %7.0f" % (rb,k[:30], m["SQ_WAVES"], wc/m["SQ_WAVES"], 100*m["SQ_ACTIVE_INST_VALU"]/wc, 100*m["SQ_WAIT_ANY"]/wc, 100*m["SQ_WAIT_INST_ANY"]/wc, m["SQ_INSTS_VALU"]/m["SQ_WAVES"], m.get("SQ_INSTS_SALU",0)/m["SQ_WAVES"], m.get("SQ_IFETCH",0)/m["SQ_WAVES"]))
PY
