#!/usr/bin/env python3
"""Turns the raw rocprofv3 output of tools/collect_profiles.sh into the small text tables kept under profiles/.

    python tools/summarize_profiles.py gpurun_out/<tag> profiles r01
"""
import collections
import csv
import json
import os
import shutil
import sys


def last_forward(rows_by_dispatch):
    """Dispatch ids of the last complete forward: from its embed_plan_kernel / seq_plan_kernel (the first launch of lamp_forward) up to the
    launch before the next one / the end of the trace."""
    ids = [k for k in rows_by_dispatch if 'lamp::' in rows_by_dispatch[k]['name']]
    first = lambda n: 'embed_plan_kernel' in n or 'seq_plan_kernel' in n   # noqa: E731 -- the first launch of lamp_forward
    emb = [i for i, k in enumerate(ids) if first(rows_by_dispatch[k]['name'])]
    return ids[emb[-2]:emb[-1]]


def load_pmc(d):
    disp = collections.OrderedDict()
    for r in csv.DictReader(open(os.path.join(d, 'p_counter_collection.csv'))):
        e = disp.setdefault(r['Dispatch_Id'], {'name': r['Kernel_Name'], 'grid': int(r['Grid_Size']),
                                               'wg': int(r['Workgroup_Size'])})
        e[r['Counter_Name']] = float(r['Counter_Value'])
    dur = {r['Dispatch_Id']: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0
           for r in csv.DictReader(open(os.path.join(d, 'p_kernel_trace.csv')))}
    return disp, dur


def kernel_only_gemm_us(src, wl):
    """GEMM kernel time per forward from a rocprofv3 --kernel-trace --stats run of bench.py (no counters): total duration
    of the gemm_nt_kernel rows / number of forwards (= embed launches)."""
    path = os.path.join(src, 'stats/p_kernel_stats.csv' if wl == 'reuters' else '../%s_stats_other/kernel_stats_%s.csv' %
                        (os.path.basename(os.path.normpath(src)), wl))
    if not os.path.exists(path):
        return None
    rows = list(csv.DictReader(open(path)))
    fwd = sum(int(r['Calls']) for r in rows if 'seq_plan_kernel' in r['Name'] or 'embed_plan_kernel' in r['Name'])
    # (the decoder chain launch is the GEMM class of the bench line too: its FLOPs are the three GEMMs it holds, its
    # time includes their two LayerNorms)
    ns = sum(float(r['TotalDurationNs']) for r in rows if 'gemm_nt_kernel' in r['Name'] or ('chain_' in r['Name'] and 'kernel' in r['Name'] and 'pack_weight' not in r['Name']))
    return ns / fwd / 1e3 if fwd else None


def short(name):
    return name.split('(')[0].replace('void lamp::', '').replace('lamp::', '')[:46]


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    cp = lambda a, b: shutil.copy(os.path.join(src, a), os.path.join(dst, '%s_%s' % (tag, b)))
    cp('bench.json', 'bench.json')
    cp('bench_under_rocprof.json', 'bench_under_rocprof.json')
    for wl in ('bibtex', 'delicious', 'synthetic4096', 'reuters_ragged', 'synthetic4096_b1024', 'synthetic4096_none',
               'driver_style_20steps', 'two_ranks_one_gpu_gloo', 'two_ranks_one_gpu_gloo_ragged',
               'eight_ranks_one_gpu_gloo', 'eight_ranks_one_gpu_gloo_ragged'):
        if not os.path.exists(os.path.join(src, 'bench_%s.json' % wl)):
            continue
        cp('bench_%s.json' % wl, 'bench_%s.json' % wl)
    cp('gemm_tiles.txt', 'gemm_tiles.txt')
    cp('gemm_tiles_sweep.txt', 'gemm_tiles_sweep.txt')
    cp('attn_variants.txt', 'attn_variants.txt')
    cp('sparse_label_attention.txt', 'sparse_label_attention.txt')
    if os.path.exists(os.path.join(src, 'gemm_trace.txt')):
        cp('gemm_trace.txt', 'gemm_trace.txt')
    cp('stats/p_kernel_stats.csv', 'bench_kernel_stats.csv')
    for a, b in (('stats_ragged/p_kernel_stats.csv', 'bench_kernel_stats_reuters_ragged.csv'),
                 ('bench_ragged_under_rocprof.json', 'bench_ragged_under_rocprof.json'),
                 ('eval_epoch_end_to_end.json', 'eval_epoch_end_to_end.json'),
                 ('rccl_control_plane_one_rank.txt', 'rccl_control_plane_one_rank.txt')):
        if os.path.exists(os.path.join(src, a)):
            cp(a, b)
    other = os.path.join(src, '..', '%s_stats_other' % os.path.basename(os.path.normpath(src)))
    for wl in ('bibtex', 'delicious', 'synthetic4096'):
        f = os.path.join(other, 'kernel_stats_%s.csv' % wl)
        if os.path.exists(f):
            shutil.copy(f, os.path.join(dst, '%s_bench_kernel_stats_%s.csv' % (tag, wl)))
    for f in ('train_reuters.json', 'train_reuters_cpu_oracle.json', 'train_delicious.json', 'gemm_gen.txt'):
        if os.path.exists(os.path.join(src, f)):
            cp(f, f)
    if os.path.exists(os.path.join(src, 'train_stats/p_kernel_stats.csv')):
        cp('train_stats/p_kernel_stats.csv', 'train_kernel_stats.csv')

    disp, dur = load_pmc(os.path.join(src, 'pmc_sq'))
    L = ['# One forward (reuters, batch 32) under rocprofv3 --pmc (SQ counters; kernels run serialized and ~4 % slower under the profiler).',
         '# clock_GHz  = SQ_BUSY_CYCLES / 32 shader engines / duration   (the sustained clock under this load, not the 2.4 GHz spec)',
         '# mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x duration x clock)  -- fraction of the matrix pipes\' cycles spent in MFMAs',
         '#              (the busy-cycle count equals 32 cycles x the algorithmic number of 16x16x4 MFMAs in the GEMMs, 64 x the 32x32x2 count in attention: no wasted matrix work)',
         '# wait_any   = SQ_WAIT_ANY / SQ_WAVE_CYCLES (waves parked on s_waitcnt / barriers);  wait_mfma = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES (issue stalls, mostly the busy matrix pipe)',
         '%-46s %8s %8s %9s %9s %9s %9s' % ('kernel', 'WGs', 'us', 'clock_GHz', 'mfma_busy', 'wait_any', 'wait_mfma')]
    for k in last_forward(disp):
        d, us = disp[k], dur[k]
        clk = d['SQ_BUSY_CYCLES'] / 32 / us / 1000
        mf = d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * us * 1000 * clk) if clk > 0 else 0
        wc = d['SQ_WAVE_CYCLES'] or 1
        L.append('%-46s %8d %8.1f %9.2f %9.2f %9.2f %9.2f' % (short(d['name']), d['grid'] // d['wg'], us, clk, mf,
                                                             d['SQ_WAIT_ANY'] / wc, d['SQ_WAIT_INST_ANY'] / wc))
    open(os.path.join(dst, '%s_mfma_busy.txt' % tag), 'w').write('\n'.join(L) + '\n')

    # fingerprint of the kernel sources the box ran (written there by tools/collect_profiles.sh)
    fp_path = os.path.join(src, 'csrc_fingerprint.txt')
    fingerprint = open(fp_path).read().strip() if os.path.exists(fp_path) else None
    traffic = {}
    tables = []
    for wl in ('reuters', 'bibtex', 'delicious'):
        res = {}
        ok = True
        for c in ('FETCH_SIZE', 'WRITE_SIZE'):
            d = os.path.join(src, 'pmc_%s_%s' % (c, wl))
            if not os.path.exists(os.path.join(d, 'p_counter_collection.csv')):
                ok = False
                break
            disp, _ = load_pmc(d)
            res[c] = [(disp[k]['name'], disp[k]['grid'], disp[k].get(c, 0.0)) for k in last_forward(disp)]
        if not ok:
            continue
        L = ['# HBM-side traffic of ONE forward (%s, batch 32), rocprofv3 PMC, separate passes for FETCH_SIZE and WRITE_SIZE.' % wl,
             '# The counters are in KiB; per the MI355X guide FETCH_SIZE under-reports wide coalesced reads by exactly 2x on gfx950,',
             '# so "fetch MB(x2)" doubles it.  These are the L2s\' memory-side requests: Infinity-Cache hits are included.',
             '%-46s %10s %12s %12s' % ('kernel', 'grid', 'fetch MB(x2)', 'write MB')]
        tf = tw = gf = gw = 0.0
        ng = 0
        for (n, g, f), (_, _, w) in zip(res['FETCH_SIZE'], res['WRITE_SIZE']):
            fm, wm = f * 1024 * 2 / 1e6, w * 1024 / 1e6
            tf, tw = tf + fm, tw + wm
            if 'gemm_nt_kernel' in n or ('chain_' in n and 'kernel' in n and 'pack_weight' not in n):
                gf, gw, ng = gf + fm, gw + wm, ng + 1
            L.append('%-46s %10d %12.2f %12.2f' % (short(n), g, fm, wm))
        L.append('%-46s %10s %12.2f %12.2f' % ('TOTAL per forward', '', tf, tw))
        L.append('%-46s %10d %12.2f %12.2f' % ('GEMM launches per forward', ng, gf, gw))
        tables.append('\n'.join(L))
        traffic[wl] = {'batch': 32, 'gemm_launches': ng, 'gemm_fetch_bytes': gf * 1e6, 'gemm_write_bytes': gw * 1e6,
                       'forward_fetch_bytes': tf * 1e6, 'forward_write_bytes': tw * 1e6,
                       'csrc_fingerprint': fingerprint, 'gemm_kernel_only_us_per_step': kernel_only_gemm_us(src, wl),
                       'source': 'profiles/%s_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; '
                                 'FETCH_SIZE x2 per the gfx950 calibration of MI355X_MICROARCH.md)' % tag}
    if tables:
        open(os.path.join(dst, '%s_hbm_traffic.txt' % tag), 'w').write('\n\n'.join(tables) + '\n')
        json.dump(traffic, open(os.path.join(dst, 'hbm_traffic.json'), 'w'), indent=1)
    print('wrote', sorted(f for f in os.listdir(dst) if f.startswith(tag)))


if __name__ == '__main__':
    main()
