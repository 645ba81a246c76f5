"""Prints DESIGN.md section 5's tables from profiles/<round>_* (the bench lines, kernel stats and PMC summary of
scripts/r0N_profiles.sh).  usage: python scripts/make_design_table.py [r05]"""
import csv
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else 'r05'
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'profiles')
FL = {'conv1a': 2.0808, 'conv2a': 22.1962, 'conv3a': 11.0981, 'conv3b': 22.1962, 'conv4a': 5.5490, 'conv4b': 11.0981,
      'conv5a': 1.3873, 'conv5b': 1.3873}                      # GFLOP per window


def load(name):
    with open(os.path.join(P, '%s_%s' % (R, name))) as fh:
        return json.load(fh)


def stats(name):
    with open(os.path.join(P, '%s_%s' % (R, name))) as fh:
        return list(csv.DictReader(fh))


e2e = load('bench_e2e.json')
pmc = load('pmc_summary.json')
n = e2e['config']['frames_per_step_per_gpu']
print('**Headline (the driver\'s command, `%s_bench_e2e.json`; this box: %.2f ms per %d frames).**  %.0f frames/s, %.2f PFLOP/s '
      'algorithmic = %.3f of the 2.5 PFLOP/s bf16 peak over the whole step; CPU baseline (oracle/torch_ref.py, %d threads) %.1f frames/s.\n'
      % (R, e2e['ms_per_step'], n, e2e['value'], e2e['algorithmic_tflops'] / 1e3, e2e['algorithmic_tflops'] / 2500.0,
         e2e['cpu_baseline']['cores'], e2e['cpu_baseline']['value']))
print('| layer | ms per %d windows | PFLOP/s (of 2.5) | matrix-pipe duty | clock GHz | traffic beyond L2, GB (algorithmic in + out) |' % n)
print('|---|---|---|---|---|---|')
# pmc entries by layer: match by kernel name fragments
keys = {'conv1a': 'conv1a_pool', 'conv2a': '<64, 128, 56, 16', 'conv3a': '<128, 256, 28, 8', 'conv3b': '<256, 256, 28, 8',
        'conv4a': 'conv_patch14_bf16_kernel<256', 'conv4b': 'conv_patch14_bf16_kernel<512', 'conv5a': 'conv_patch7_bf16_kernel<0',
        'conv5b': 'conv_patch7_bf16_kernel<1'}
ALG = {'conv1a': 2.408 + 6.423, 'conv2a': 6.423 + 1.606, 'conv3a': 1.606 + 3.211, 'conv3b': 3.211 + 0.401, 'conv4a': 0.401 + 0.803,
       'conv4b': 0.803 + 0.100, 'conv5a': 0.100 + 0.100, 'conv5b': 0.100 + 0.100}      # MB per window: input + output images
for layer, ms in e2e['stage_ms_per_step'].items():
    if layer not in FL:
        continue
    pf = FL[layer] * n / ms / 1e3
    ent = [v for k, v in pmc.items() if keys[layer] in k and 'mfma_pipe_duty' in v]
    duty = clk = hbm = None
    if ent:
        v = ent[0]
        duty = v.get('mfma_pipe_duty')
        clk = v.get('GRBM_GUI_ACTIVE', 0) / 8 / (v['ms_mean_under_pmc'] * 1e-3) / 1e9
        hbm = v.get('hbm_bytes_per_launch', 0) / 1e9
    print('| %s | %.2f | %.2f (%.3f) | %s | %s | %s (%.2f) |' % (layer, ms, pf, pf / 2.5, '%.2f' % duty if duty else '-',
                                                              '%.2f' % clk if clk else '-', '%.2f' % hbm if hbm else '-', ALG[layer] * n / 1e3))
c1 = e2e['stage_ms_per_step']['conv1a']
print('| conv1a against its own roof (HBM): %.2f GB algorithmic / %.2f ms = %.2f TB/s = %.2f of 8 TB/s | | | | | |' % (ALG['conv1a'] * n / 1e3, c1, ALG['conv1a'] * n / 1e3 / c1, ALG['conv1a'] * n / 1e3 / c1 / 8.0))
head = {k: v for k, v in e2e['stage_ms_per_step'].items() if k not in FL and k != 'video_prep'}
print('| head (proj + W.x + ConvGRU + folded head + softmax) | %.2f | | | | |' % sum(head.values()))
print()
cfg = load('configs.json')
print('| configuration (BASELINE.json) | shape per GPU | ms | frames/s |')
print('|---|---|---|---|')
rows = [('1 frame-wise ShallowNet, forward / training step', '512 frames of 112x112', 'cfg1_shallownet_112_n512_fwd', 'cfg1_shallownet_112_n512_fwd_bwd'),
        ('2 fc-GRU f32, forward / training step', '64 clips x T = 16', 'cfg2_fcgru_f32_B64_T16_fwd', 'cfg2_fcgru_f32_B64_T16_train_step'),
        ('3 gaze_grcn head on features, forward / training step', '64 x 16', 'cfg3_grcn_bf16_B64_T16_head_fwd', 'cfg3_grcn_bf16_B64_T16_head_train_step'),
        ('3 end to end (the headline) / end-to-end fine-tune step', '64 x 16 = 1024 windows / 16 x 16 = 256', 'cfg3_grcn_bf16_B64_T16_e2e_fwd', 'cfg3_grcn_bf16_B16_T16_end_to_end_finetune_step'),
        ('4 head training step', '8 x 35', 'cfg4_grcn_bf16_B8_T35_train_step', None),
        ('5 cascade forward / forward + backward', '16 x 35', 'cfg5_cascade_bf16_B16_T35_fwd', 'cfg5_cascade_bf16_B16_T35_fwd_bwd'),
        ('5 conv stack + cascade, joint training step', '16 x 35 = 560 windows', 'cfg5_c3d_finetune_plus_cascade_B16_T35_train_step', None),
        ('reference shapes: training forward / step; extraction forward', '28 x 42; 14 x 105', 'ref_train_B28_T42_grcn_bf16_head_fwd', 'ref_train_B28_T42_grcn_bf16_head_train_step')]
for label, shape, a, b in rows:
    ms = '%.3f' % cfg[a]['ms'] + (' / %.3f' % cfg[b]['ms'] if b else '')
    fs = '%.0f' % cfg[a]['frames_per_s'] + (' / %.0f' % cfg[b]['frames_per_s'] if b else '')
    if a.startswith('ref_train'):
        ms += '; %.3f' % cfg['ref_extract_map_B14_T105_grcn_bf16_head_fwd']['ms']
        fs += '; %.0f' % cfg['ref_extract_map_B14_T105_grcn_bf16_head_fwd']['frames_per_s']
    print('| %s | %s | %s | %s |' % (label, shape, ms, fs))
print()
ft = stats('finetune_B16_T16_kernel_stats.csv')
print('Fine-tune step, kernels over 0.7 ms (`%s_finetune_B16_T16_kernel_stats.csv`, 256 windows, under rocprofv3):\n' % R)
print('| kernel | ms | PFLOP/s (of 2.5) |')
print('|---|---|---|')
want = [('conv_patch_bf16_kernel<128, 64, 56, 16, false, false, true, true>', 'conv2a input gradient (dense)', 'conv2a'),
        ('wgrad_patch_bf16_kernel<256, 256, 28, 8>', 'conv3b filter gradient', 'conv3b'),
        ('conv_patch_bf16_kernel<64, 128, 56, 16, true, true, false, false>', 'conv2a forward (arg-max)', 'conv2a'),
        ('wgrad_patch_bf16_kernel<64, 128, 56, 16>', 'conv2a filter gradient', 'conv2a'),
        ('conv_patch_bf16_kernel<256, 256, 28, 8, true, true, false, false>', 'conv3b forward (arg-max)', 'conv3b'),
        ('conv_patch_bf16_kernel<256, 256, 28, 8, false, false, true, false>', 'conv3b input gradient (masked)', 'conv3b'),
        ('wgrad_patch_bf16_kernel<128, 256, 28, 8>', 'conv3a filter gradient (+ bias)', 'conv3a'),
        ('conv_patch_bf16_kernel<256, 128, 28, 8, false, false, true, true>', 'conv3a input gradient (dense)', 'conv3a'),
        ('wgrad_patch_bf16_kernel<512, 512, 14, 4>', 'conv4b filter gradient (window pairs)', 'conv4b'),
        ('conv_patch_bf16_kernel<128, 256, 28, 8, false, false, false, false>', 'conv3a forward', 'conv3a'),
        ('conv_patch14_bf16_kernel<512, true, true, false, 512, false>', 'conv4b forward (arg-max)', 'conv4b'),
        ('conv_patch14_bf16_kernel<512, false, false, true, 512, false>', 'conv4b input gradient', 'conv4b'),
        ('conv1a_pool_bf16_kernel<false, true, 0>', 'conv1a forward (arg-max)', 'conv1a'),
        ('conv1a_wgrad_bf16_kernel', 'conv1a filter gradient', 'conv1a'),
        ('wgrad_patch_bf16_kernel<256, 512, 14, 4>', 'conv4a filter gradient (window pairs, + bias)', 'conv4a'),
        ('unpool8_rows_kernel<128>', 'un-pool into conv2a', None),
        ('conv_patch14_bf16_kernel<512, false, false, true, 256, true>', 'conv4a input gradient (dense)', 'conv4a'),
        ('conv_patch14_bf16_kernel<256, false, false, false, 512, false>', 'conv4a forward', 'conv4a')]
for frag, label, layer in want:
    r = [x for x in ft if frag in x['Name']]
    if not r:
        continue
    ms = float(r[0]['AverageNs']) / 1e6
    pf = FL[layer] * 256 / ms / 1e3 if layer else None
    print('| %s | %.2f | %s |' % (label, ms, '%.2f (%.2f)' % (pf, pf / 2.5) if pf else 'HBM: 5.1 TB/s'))
