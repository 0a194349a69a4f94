"""State-dict specification of the reference `nets.Achelous.Achelous` (key names, shapes, parameter/buffer kind),
generated from the architecture description alone.

This is what makes `achelous_amd.Achelous` a drop-in for `load_state_dict(torch.load(path))` (achelous.py:171):
the same 771 / 825 / 920 keys as the reference for EN-S0 / EN-S2 / MV-S2, in the same order.  Checked against the
key lists captured from the imported reference (tests/golden/*.keys.json) by tests/test_abi_and_host.py.

Each entry: (key, shape, kind) with kind in {'param', 'buffer', 'buffer_i64'}.
"""
import math

WIDTHS = {'S0': [32, 48, 96, 176], 'S1': [32, 48, 120, 224], 'S2': [32, 64, 144, 288]}   # neck/ghostdualfpn.py:20-25

EDGENEXT = {   # backbone/vision/edgenext_modules/model.py:14-66 ; heads default 8 (edgenext.py:14)
    'S0': dict(depths=[2, 2, 6, 2], dims=[32, 48, 96, 176], heads=4, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
    'S1': dict(depths=[3, 3, 9, 3], dims=[32, 48, 120, 224], heads=4, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
    'S2': dict(depths=[3, 3, 9, 3], dims=[32, 64, 144, 288], heads=8, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
}
MOBILEVIT = {  # backbone/vision/mobilevit_modules/mobilevit.py:225-240
    'S0': dict(dims=[64, 80, 96], ch=[16, 16, 32, 32, 48, 48, 96, 96, 96, 96, 176], exp=2),
    'S1': dict(dims=[96, 120, 144], ch=[16, 32, 32, 32, 48, 48, 120, 120, 120, 120, 224], exp=4),
    'S2': dict(dims=[144, 192, 240], ch=[16, 32, 32, 32, 64, 64, 144, 144, 144, 144, 288], exp=4),
}


class _Spec(list):
    def p(self, key, *shape):
        self.append((key, tuple(shape), 'param'))

    def wb(self, pfx, wshape, bias=True):
        self.p(pfx + '.weight', *wshape)
        if bias:
            self.p(pfx + '.bias', wshape[0])

    def bn(self, pfx, c):
        self.p(pfx + '.weight', c)
        self.p(pfx + '.bias', c)
        self.append((pfx + '.running_mean', (c,), 'buffer'))
        self.append((pfx + '.running_var', (c,), 'buffer'))
        self.append((pfx + '.num_batches_tracked', (), 'buffer_i64'))

    def ln(self, pfx, c):
        self.p(pfx + '.weight', c)
        self.p(pfx + '.bias', c)


def _stn(s, pfx, cin, k):                       # pointnet_utils.py:10-85
    s.wb(pfx + '.conv1', (64, cin, 1))
    s.wb(pfx + '.conv2', (128, 64, 1))
    s.wb(pfx + '.conv3', (1024, 128, 1))
    s.wb(pfx + '.fc1', (512, 1024))
    s.wb(pfx + '.fc2', (256, 512))
    s.wb(pfx + '.fc3', (k * k, 256))
    for i, c in enumerate((64, 128, 1024, 512, 256), 1):
        s.bn(f'{pfx}.bn{i}', c)


def _pointnet(s, pc_channels, pc_classes):      # pointnet_sem_seg.py:13-24, pointnet_utils.py:88-101
    p = 'pc_seg_model'
    _stn(s, p + '.feat.stn', pc_channels, 3)
    s.wb(p + '.feat.conv1', (32, pc_channels, 1))
    s.wb(p + '.feat.conv2', (64, 32, 1))
    s.wb(p + '.feat.conv3', (128, 64, 1))
    for i, c in enumerate((32, 64, 128), 1):
        s.bn(f'{p}.feat.bn{i}', c)
    _stn(s, p + '.feat.fstn', 32, 32)
    s.wb(p + '.conv1', (128, 160, 1))
    s.wb(p + '.conv2', (100, 128, 1))
    s.wb(p + '.conv3', (64, 100, 1))
    s.wb(p + '.conv4', (pc_classes, 64, 1))
    for i, c in enumerate((128, 100, 64), 1):
        s.bn(f'{p}.bn{i}', c)


# PointNet++ branch (`pc_seg='pn2'`): OUR OWN specification - the reference snapshot has no PointNet++ code (SURVEY.md top;
# nets/Achelous.py:31-32 builds only 'pn').  Structure and key names follow the published single-scale-grouping semantic
# segmentation model of the public Pointnet_Pointnet2_pytorch project (sa1-4 / fp4-1 / conv1 / bn1 / conv2, with its channel
# widths), re-sized to the 512-point clouds of this path: level k keeps N / div points, 32 samples per ball, radii in the units
# of the column-normalised cloud (achelous.py:240).  Geometry rules (FPS start, distance form, tie-breaks): DESIGN.md section 5b.
# (Round 4, profiles/scripts/pn2_spec_search.py: the reference's one PointNet++ datum — README.md:81,83: +0.09 M parameters, +0.08 GFLOPs over PointNet,
#  i.e. ~2.01 M / ~251 M MACs — is fitted by the MULTI-scale model of that project (1.88 M; 190-345 M MACs depending on level sizes), not by this
#  single-scale one (0.97 M, 207 M MACs).  The single-scale variant stays the specification this round: radii and sample counts of the multi-scale one
#  would be as much our own choice, and the branch is self-specified either way; DESIGN.md 5b has the table.)
PN2 = dict(
    sa=[dict(div=2, radius=0.03, nsample=32, mlp=[32, 32, 64]),
        dict(div=8, radius=0.06, nsample=32, mlp=[64, 64, 128]),
        dict(div=32, radius=0.12, nsample=32, mlp=[128, 128, 256]),
        dict(div=128, radius=0.24, nsample=32, mlp=[256, 256, 512])],
    fp=[[256, 256], [256, 256], [256, 128], [128, 128, 128]],          # fp4, fp3, fp2, fp1
    head=128,
)


# The MULTI-scale-grouping variant (`pc_seg='pn2_msg'`, round 6): the reference's only PointNet++ datum — README.md:81,83: +0.09 M parameters and
# +0.08 GFLOPs over the PointNet model — is fitted by the multi-scale semantic-segmentation model of the same public project (two radii per level, two shared-MLP stacks whose
# maxima are concatenated), not by the single-scale one above (profiles/scripts/pn2_spec_search.py; DESIGN.md 5b has the parameter / MAC table).  Widths, sample counts and key
# names (`conv_blocks.<scale>.<layer>`, `bn_blocks...`) are that model's; level sizes and the radius ladder are PN2's (each level: [r, 2r]); every tie / distance rule is
# PN2's, and a grouped row is [xyz - centroid | features] at BOTH scales (the published multi-scale code concatenates the other way round: a permutation of the first
# layer's input columns, fixed here so that one grouping kernel serves both variants).  Self-specified and self-checked like PN2: PARITY UNPINNED.
PN2_MSG = dict(
    sa=[dict(div=2, scales=[dict(radius=0.03, nsample=16, mlp=[16, 16, 32]), dict(radius=0.06, nsample=32, mlp=[32, 32, 64])]),
        dict(div=8, scales=[dict(radius=0.06, nsample=16, mlp=[64, 64, 128]), dict(radius=0.12, nsample=32, mlp=[64, 96, 128])]),
        dict(div=32, scales=[dict(radius=0.12, nsample=16, mlp=[128, 196, 256]), dict(radius=0.24, nsample=32, mlp=[128, 196, 256])]),
        dict(div=128, scales=[dict(radius=0.24, nsample=16, mlp=[256, 256, 512]), dict(radius=0.48, nsample=32, mlp=[256, 384, 512])])],
    fp=[[256, 256], [256, 256], [256, 128], [128, 128, 128]],          # fp4, fp3, fp2, fp1
    head=128,
)
PN2_VARIANTS = {'pn2': PN2, 'pn2_msg': PN2_MSG}


def pn2_scales(level):
    """The (radius, nsample, mlp) stacks of a set-abstraction level: one for the single-scale specification, two for the multi-scale one."""
    return level['scales'] if 'scales' in level else [dict(radius=level['radius'], nsample=level['nsample'], mlp=level['mlp'])]


def pn2_level_width(level):
    return sum(sc['mlp'][-1] for sc in pn2_scales(level))


def _pointnet2(s, pc_channels, pc_classes, spec=None):
    spec = spec or PN2
    p = 'pc_seg_model'
    feat = [pc_channels] + [pn2_level_width(lvl) for lvl in spec['sa']]        # feature width at l0 .. l4
    for k, lvl in enumerate(spec['sa']):
        if 'scales' not in lvl:
            cin = feat[k] + 3
            for i, c in enumerate(lvl['mlp']):
                s.wb(f'{p}.sa{k + 1}.mlp_convs.{i}', (c, cin, 1, 1))
                cin = c
            for i, c in enumerate(lvl['mlp']):
                s.bn(f'{p}.sa{k + 1}.mlp_bns.{i}', c)
            continue
        for j, sc in enumerate(lvl['scales']):
            cin = feat[k] + 3
            for i, c in enumerate(sc['mlp']):
                s.wb(f'{p}.sa{k + 1}.conv_blocks.{j}.{i}', (c, cin, 1, 1))
                cin = c
            for i, c in enumerate(sc['mlp']):
                s.bn(f'{p}.sa{k + 1}.bn_blocks.{j}.{i}', c)
    cur = feat[-1]
    L = len(spec['sa'])
    for j, widths in enumerate(spec['fp']):
        lvl = L - 1 - j                                                  # dense level of this propagation
        cin = cur + (feat[lvl] if lvl > 0 else 0)
        for i, c in enumerate(widths):
            s.wb(f'{p}.fp{lvl + 1}.mlp_convs.{i}', (c, cin, 1))
            cin = c
        for i, c in enumerate(widths):
            s.bn(f'{p}.fp{lvl + 1}.mlp_bns.{i}', c)
        cur = widths[-1]
    s.wb(p + '.conv1', (spec['head'], cur, 1))
    s.bn(p + '.bn1', spec['head'])
    s.wb(p + '.conv2', (pc_classes, spec['head'], 1))


def _edgenext(s, pfx, phi):                     # edgenext.py:9-62
    cfg = EDGENEXT[phi]
    dims = cfg['dims']
    s.wb(pfx + '.downsample_layers.0.0', (dims[0], 3, 4, 4))
    s.ln(pfx + '.downsample_layers.0.1', dims[0])
    for i in range(3):
        s.ln(f'{pfx}.downsample_layers.{i + 1}.0', dims[i])
        s.wb(f'{pfx}.downsample_layers.{i + 1}.1', (dims[i + 1], dims[i], 2, 2))
    for i in range(4):
        d = dims[i]
        for j in range(cfg['depths'][i]):
            b = f'{pfx}.stages.{i}.{j}'
            if i > 0 and j == cfg['depths'][i] - 1:                 # SDTAEncoder (sdta_encoder.py:8-37)
                sc = cfg['scales'][i]
                width = max(int(math.ceil(d / sc)), int(math.floor(d // sc)))
                s.p(b + '.gamma_xca', d)
                s.p(b + '.gamma', d)
                for n in range(sc - 1):
                    s.wb(f'{b}.convs.{n}', (width, 1, 3, 3))
                if i == 1:                                          # use_pos_embd_xca=[False, True, False, False]
                    s.wb(b + '.pos_embd.token_projection', (d, 64, 1, 1))
                s.ln(b + '.norm_xca', d)
                s.p(b + '.xca.temperature', cfg['heads'], 1, 1)
                s.wb(b + '.xca.qkv', (3 * d, d))
                s.wb(b + '.xca.proj', (d, d))
            else:                                                   # ConvEncoder (conv_encoder.py:7-17)
                k = cfg['ks'][i]
                s.p(b + '.gamma', d)
                s.wb(b + '.dwconv', (d, 1, k, k))
            s.ln(b + '.norm', d)
            s.wb(b + '.pwconv1', (4 * d, d))
            s.wb(b + '.pwconv2', (d, 4 * d))
    s.ln(pfx + '.norm', dims[-1])                                   # unused classifier tail (edgenext.py:56-57)
    s.wb(pfx + '.head', (1000, dims[-1]))


def _conv_bn_seq(s, pfx, cout, cin, k, groups=1):                   # nn.Sequential(conv(bias=False), BN, act)
    s.wb(pfx + '.0', (cout, cin // groups, k, k), bias=False)
    s.bn(pfx + '.1', cout)


def _mobilevit(s, pfx, phi):                    # mobilevit.py:168-196
    cfg = MOBILEVIT[phi]
    ch, dims, exp = cfg['ch'], cfg['dims'], cfg['exp']
    _conv_bn_seq(s, pfx + '.conv1', ch[0], 3, 3)
    mv2 = [(ch[0], ch[1]), (ch[1], ch[2]), (ch[2], ch[3]), (ch[2], ch[3]), (ch[3], ch[4]), (ch[5], ch[6]), (ch[7], ch[8])]
    for i, (inp, oup) in enumerate(mv2):
        hid = int(inp * exp)
        b = f'{pfx}.mv2.{i}.conv'
        s.wb(b + '.0', (hid, inp, 1, 1), bias=False)
        s.bn(b + '.1', hid)
        s.wb(b + '.3', (hid, 1, 3, 3), bias=False)
        s.bn(b + '.4', hid)
        s.wb(b + '.6', (oup, hid, 1, 1), bias=False)
        s.bn(b + '.7', oup)
    L = [2, 4, 3]
    mlp = [int(dims[0] * 2), int(dims[1] * 4), int(dims[2] * 4)]
    chan = [ch[5], ch[7], ch[9]]
    for i in range(3):
        b = f'{pfx}.mvit.{i}'
        d, c = dims[i], chan[i]
        _conv_bn_seq(s, b + '.conv1', c, c, 3)
        _conv_bn_seq(s, b + '.conv2', d, c, 1)
        for l in range(L[i]):
            a = f'{b}.transformer.layers.{l}.0'
            s.ln(a + '.norm', d)
            s.wb(a + '.fn.to_qkv', (96, d), bias=False)             # heads 4 x dim_head 8 (mobilevit.py:142)
            s.wb(a + '.fn.to_out.0', (d, 32))
            f = f'{b}.transformer.layers.{l}.1'
            s.ln(f + '.norm', d)
            s.wb(f + '.fn.net.0', (mlp[i], d))
            s.wb(f + '.fn.net.3', (d, mlp[i]))
        _conv_bn_seq(s, b + '.conv3', c, d, 1)
        _conv_bn_seq(s, b + '.conv4', c, 2 * c, 3)
    _conv_bn_seq(s, pfx + '.conv2', ch[-1], ch[-2], 1)


def _ghost(s, pfx, inp, oup):                   # ghost_conv.py:6-23
    init = math.ceil(oup / 2)
    _conv_bn_seq(s, pfx + '.primary_conv', init, inp, 1)
    _conv_bn_seq(s, pfx + '.cheap_operation', init, init, 3, groups=init)


def _ghost_bottleneck(s, pfx, inp, mid, out):   # ghost_conv.py:32-56 (stride 1, in != out)
    _ghost(s, pfx + '.ghost1', inp, mid)
    _ghost(s, pfx + '.ghost2', mid, out)
    s.wb(pfx + '.shortcut.0', (inp, 1, 3, 3), bias=False)
    s.bn(pfx + '.shortcut.1', inp)
    s.wb(pfx + '.shortcut.2', (out, inp, 1, 1), bias=False)
    s.bn(pfx + '.shortcut.3', out)


def _baseconv(s, pfx, cin, cout, k=1):          # normal_conv.py:36-47
    s.wb(pfx + '.conv', (cout, cin, k, k), bias=False)
    s.bn(pfx + '.bn', cout)


def _csp_bottleneck(s, pfx, cin, cout, expansion=0.5):              # cspdualfpn.py:42-57
    hidden = int(cout * expansion)
    _baseconv(s, pfx + '.conv1', cin, hidden)
    _baseconv(s, pfx + '.conv2', hidden, cout, 3)


def _csp_layer(s, pfx, cin, cout):                                  # cspdualfpn.py:60-78, n = 1
    hidden = int(cout * 0.5)
    _baseconv(s, pfx + '.conv1', cin, hidden)
    _baseconv(s, pfx + '.conv2', cin, hidden)
    _baseconv(s, pfx + '.conv3', 2 * hidden, cout)
    _csp_bottleneck(s, pfx + '.m.0', hidden, hidden, 1.0)


def _neck_csp(s, phi, backbone, num_seg):       # cspdualfpn.py:81-191
    f = 'image_radar_encoder.fpn'
    w = WIDTHS[phi]
    if backbone == 'en':
        _edgenext(s, f + '.backbone', phi)
    else:
        _mobilevit(s, f + '.backbone', phi)
    c_ = w[3] // 2
    _baseconv(s, f + '.spp.cv1', w[3], c_)
    _baseconv(s, f + '.spp.cv2', 4 * c_, w[3])
    _baseconv(s, f + '.upsample_5_to_4.upsample.0', w[3], w[2])
    _csp_layer(s, f + '.ghost_5_to_4', 2 * w[2], w[2])
    _baseconv(s, f + '.upsample_4_to_3.upsample.0', w[2], w[1])
    _csp_layer(s, f + '.ghost_4_to_3', 2 * w[1], w[1])
    for sa in ('stage_3_lane_seg', 'stage_3_semantic_seg'):
        c = w[1] // 8
        for nm in ('cweight', 'cbias', 'sweight', 'sbias'):
            s.p(f'{f}.{sa}.{nm}', 1, c, 1, 1)
        s.ln(f'{f}.{sa}.gn', c)
    for name, oup in (('lane', 2), ('se', num_seg)):
        for lvl, cin, cout in (('3_to_2', w[1], w[1]), ('2_to_1', w[1], w[0]), ('1_to_0', w[0], w[0])):
            _baseconv(s, f'{f}.{name}_seg_{lvl}.upsample.0', cin, cout)
            _csp_bottleneck(s, f'{f}.{name}_seg_ghost_{lvl}', cout, cout)
        _csp_bottleneck(s, f'{f}.{name}_seg_head', w[0], oup)


def _neck(s, phi, backbone, num_seg):           # ghostdualfpn.py:42-152
    f = 'image_radar_encoder.fpn'
    w = WIDTHS[phi]
    if backbone == 'en':
        _edgenext(s, f + '.backbone', phi)
    else:
        _mobilevit(s, f + '.backbone', phi)
    c_ = w[3] // 2
    _baseconv(s, f + '.spp.cv1', w[3], c_)
    _baseconv(s, f + '.spp.cv2', 4 * c_, w[3])
    _baseconv(s, f + '.upsample_5_to_4.upsample.0', w[3], w[2])
    _ghost_bottleneck(s, f + '.ghost_5_to_4', 2 * w[2], 2 * w[2], w[2])
    _baseconv(s, f + '.upsample_4_to_3.upsample.0', w[2], w[1])
    _ghost_bottleneck(s, f + '.ghost_4_to_3', 2 * w[1], 2 * w[1], w[1])
    for sa in ('stage_3_lane_seg', 'stage_3_semantic_seg'):         # ShuffleAttention G=4 (shuffle_attention.py:9-19)
        c = w[1] // 8
        for nm in ('cweight', 'cbias', 'sweight', 'sbias'):
            s.p(f'{f}.{sa}.{nm}', 1, c, 1, 1)
        s.ln(f'{f}.{sa}.gn', c)
    for name, oup in (('lane', 2), ('se', num_seg)):
        for lvl, cin, cout in (('3_to_2', w[1], w[1]), ('2_to_1', w[1], w[0]), ('1_to_0', w[0], w[0])):
            _baseconv(s, f'{f}.{name}_seg_{lvl}.upsample.0', cin, cout)
            _ghost(s, f'{f}.{name}_seg_ghost_{lvl}', cout, cout)
        _ghost(s, f'{f}.{name}_seg_head', w[0], oup)


def _radar(s, phi, radar_channels):             # RadarEncoder.py:44-97, dcn.py:6-47
    w = WIDTHS[phi]
    ch = [radar_channels, w[0] // 4, w[0] // 4, w[0] // 4, w[1] // 4, w[1] // 4, w[2] // 4, w[2] // 4, w[3] // 4]
    down = [True, True, False, True, False, True, False, True]
    for i in range(8):
        b = f'image_radar_encoder.radar_encoder.rc_blocks.{i}'
        c, co = ch[i], ch[i + 1]
        d = b + '.radar_conv.deformable_conv'
        s.wb(d + '.offset_conv', (18, c, 3, 3))
        s.wb(d + '.modulator_conv', (9, c, 3, 3))
        s.wb(d + '.regular_conv', (c, c, 3, 3), bias=False)
        s.wb(b + '.weight_conv1', (c, c, 1, 1))
        s.bn(b + '.norm', c)
        k = 3 if down[i] else 1
        s.wb(b + '.weight_conv2', (co, c, k, k))


def _eca_k(channel, b=1, gamma=2):              # eca.py:8-10
    k = int(abs((math.log(channel, 2) + b) / gamma))
    return k if k % 2 else k + 1


def _fusion(s, phi):                            # IREncoder.py:46-69
    w = WIDTHS[phi]
    for stage, c in ((3, w[1]), (4, w[2]), (5, w[3])):
        e = 'image_radar_encoder'
        s.p(f'{e}.channel_attn_stage{stage}.0.conv.weight', 1, 1, _eca_k(c))
        s.p(f'{e}.channel_attn_stage{stage}.1.conv.weight', 1, 1, _eca_k(c // 4))
        s.bn(f'{e}.norm_stage{stage}', c * 5 // 4)


def _head(s, phi, num_det, nano_head):          # decouplehead.py:16-56
    w = WIDTHS[phi]
    base = 64 if nano_head else 256
    ins = [c * 5 // 4 for c in w[1:]]

    def dw(pfx):
        s.wb(pfx + '.conv.dconv', (base, 1, 5, 5), bias=False)
        s.wb(pfx + '.conv.pconv', (base, base, 1, 1), bias=False)
        s.bn(pfx + '.bn', base)
    for group in ('cls_convs', 'reg_convs'):
        for k in range(3):
            dw(f'det_head.{group}.{k}.0')
            dw(f'det_head.{group}.{k}.1')
        if group == 'cls_convs':
            pass
    # registration order in the reference: cls_convs, reg_convs, cls_preds, reg_preds, obj_preds, stems
    for k in range(3):
        s.wb(f'det_head.cls_preds.{k}', (num_det, base, 1, 1))
    for k in range(3):
        s.wb(f'det_head.reg_preds.{k}', (4, base, 1, 1))
    for k in range(3):
        s.wb(f'det_head.obj_preds.{k}', (1, base, 1, 1))
    for k in range(3):
        _baseconv(s, f'det_head.stems.{k}', ins[k], base)


def state_dict_spec(num_det, num_seg, phi='S0', backbone='en', pc_channels=6, pc_classes=9, nano_head=True,
                    radar_channels=3, neck='gdf', pc_seg='pn'):
    """Ordered [(key, shape, kind)] of the reference state_dict for neck in {'gdf', 'cdf'}, pc_seg='pn'; with pc_seg='pn2' the
    `pc_seg_model.*` keys are those of our own PointNet++ specification (PN2 above)."""
    if pc_seg not in ('pn', 'pn2', 'pn2_msg', 'none'):
        raise NotImplementedError(f"pc_seg={pc_seg!r}: 'pn' (reference), 'pn2' / 'pn2_msg' (own specifications) and 'none' (Achelous3T, nets/Achelous.py:56-76) are built")
    if phi not in WIDTHS or backbone not in ('en', 'mv') or neck not in ('gdf', 'cdf'):
        raise NotImplementedError(f"backbone={backbone!r}, phi={phi!r}, neck={neck!r}: only 'en'/'mv' with S0/S1/S2 and gdf/cdf are built")
    s = _Spec()
    if pc_seg != 'none':
        if pc_seg == 'pn':
            _pointnet(s, pc_channels, pc_classes)
        else:
            _pointnet2(s, pc_channels, pc_classes, PN2_VARIANTS[pc_seg])
    (_neck if neck == 'gdf' else _neck_csp)(s, phi, backbone, num_seg)
    _radar(s, phi, radar_channels)
    _fusion(s, phi)
    _head(s, phi, num_det, nano_head)
    return list(s)
