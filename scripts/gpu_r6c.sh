#!/bin/bash
# round 6: kernel trace + MFMA / stall / traffic counters of mc_conv3x3 at the four large layer shapes (scripts/gpu_conv_prof.sh), one summary table
ulimit -c 0
for i in 1 2 3 4; do bash scripts/gpu_conv_prof.sh r6c $i > /dev/null 2>&1; done
cd $GRAFT_REPO_ROOT/gpurun_out/r6c
python - <<'PY' | tee conv_summary.txt
import csv
shapes = {1: "N=2  64-> 64  370x1226", 2: "N=2 112->112  370x1226", 3: "N=2  64-> 64 1000x1500", 4: "N=2 112->112 1000x1500"}
flops = {1: 2*2*370*1226*64*64*9, 2: 2*2*370*1226*112*112*9, 3: 2*2*1000*1500*64*64*9, 4: 2*2*1000*1500*112*112*9}
print("mc_conv3x3 (conv3x3_kernel), rocprofv3 --kernel-trace --stats and --pmc passes (scripts/gpu_conv_prof.sh); 1024 SIMDs, v_mfma_f32_32x32x2_f32 = 64 cycles each")
for i in (1, 2, 3, 4):
    ks = {r['Name']: r for r in csv.DictReader(open('conv%d_kernel_stats.csv' % i))}
    for k, r in ks.items():
        if 'conv3x3_kernel' in k:
            dur = float(r['AverageNs']); mn = float(r['MinNs'])
            print("%s  %s: avg %.1f us, min %.1f us (%.1f TFLOP/s at the min), VGPR %s AGPR %s LDS %s, grid %s threads" % (shapes[i], k.split('(')[0].replace('void mc::', ''), dur / 1e3, mn / 1e3, flops[i] / mn / 1e3, r['VGPR'], r['AGPR'], r['LDS'], r['GridX']))
        if 'conv_prep' in k:
            print("   conv_prep_kernel: avg %.1f us" % (float(r['AverageNs']) / 1e3))
    for r in csv.DictReader(open('conv%d_pmc.csv' % i)):
        if 'conv3x3_kernel' in r['Kernel']:
            g = float(r['GRBM_GUI_ACTIVE']) / 8; busy = float(r['SQ_VALU_MFMA_BUSY_CYCLES']) / 1024; wc = float(r['SQ_WAVE_CYCLES']) * 4 / 1024
            print("   cycles per XCD %.0f (clock %.2f GHz while profiled), matrix pipe busy per SIMD %.0f cycles = %.3f of the launch, MFMA instructions %.0f, "
                  "wave life %.0f cycles, waiting at s_waitcnt %.1f %% of it; HBM read %.1f MB (FETCH_SIZE as reported: dword loads), written %.1f MB"
                  % (g, g / dur, busy, busy / g, float(r['SQ_INSTS_MFMA']), wc, 100.0 * float(r['SQ_WAIT_ANY']) / float(r['SQ_WAVE_CYCLES']),
                     float(r['FETCH_SIZE']) / 1024, float(r['WRITE_SIZE']) / 1024))
PY
