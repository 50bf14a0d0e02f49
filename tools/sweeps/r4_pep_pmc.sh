#!/bin/bash
# PMC counters of the peptide fill (2M sequences of 20-40 residues): where the SIMD time goes
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4/pep_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq -- python3 $R/tools/sweeps/r4_peptides.py 2000000 $PEPOPTS > $OUT/sq.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/lds -- python3 $R/tools/sweeps/r4_peptides.py 2000000 $PEPOPTS > $OUT/lds.log 2>&1
timeout -k 10 200 rocprofv3 --pmc SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM --output-format csv -d $OUT/act -- python3 $R/tools/sweeps/r4_peptides.py 2000000 $PEPOPTS > $OUT/act.log 2>&1
python3 - <<PY
import csv, glob, collections
for sub in ("sq", "lds", "act"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            if "swg_diag_dyn" in r["Kernel_Name"] or "swg_fill_kernel" in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:56]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(sub, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
