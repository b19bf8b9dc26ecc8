#!/bin/bash
ulimit -c 0
for i in 1 3 4; do bash scripts/gpu_conv_prof.sh r6c $i > /dev/null 2>&1; done
cd $GRAFT_REPO_ROOT/gpurun_out/r6c
python - <<'PY'
import csv
for i in (1,3,4):
    ks={r['Name']:r for r in csv.DictReader(open('conv%d_kernel_stats.csv'%i))}
    for k,r in ks.items():
        if 'conv3x3_kernel' in k: dur=float(r['AverageNs']); print(i, k[:45], 'avg ns', dur, 'min', r['MinNs'], 'vgpr', r['VGPR'], r['AGPR'], 'grid', r['GridX'])
    for r in csv.DictReader(open('conv%d_pmc.csv'%i)):
        if 'conv3x3_kernel' in r['Kernel']:
            g=float(r['GRBM_GUI_ACTIVE'])/8; busy=float(r['SQ_VALU_MFMA_BUSY_CYCLES'])/1024; wc=float(r['SQ_WAVE_CYCLES'])*4/1024
            print('   cycles/XCD %.0f  clock %.2f GHz  mfma busy/SIMD %.0f (%.3f)  wave cycles %.0f  WAIT_ANY %.3f  WAIT_INST_ANY %.3f  FETCH MB %.1f WRITE MB %.1f' % (g, g/dur, busy, busy/g, wc, float(r['SQ_WAIT_ANY'])/float(r['SQ_WAVE_CYCLES']), float(r['SQ_WAIT_INST_ANY'])/float(r['SQ_WAVE_CYCLES']), float(r['FETCH_SIZE'])/1024, float(r['WRITE_SIZE'])/1024))
PY
