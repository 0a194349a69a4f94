"""`Achelous.forward` in `.train()`: the whole model on native forward / backward kernels.

The reference trains through ATen autograd (train.py:400-420, utils/utils_fit.py:37-166: forward in training mode, losses, backward).
Training mode differs from the inference path in one respect that matters — BatchNorm normalises with BATCH statistics and updates its
running estimates (DropPath / Dropout are built with rate 0: edgenext_modules/model.py:14-30, sdta_encoder.py:10-11) — and that is
exactly what the inference engine cannot do, because it folds the running statistics into the convolutions.  So this is a second,
unfused statement of the same network: a functional evaluator over the module's parameter tree (reference state-dict keys), every
arithmetic operation a `torch.autograd.Function` from train_functional.py / train_ops.py whose forward and backward are hand-written HIP
kernels.  What torch does here is tensor plumbing: views, `cat` / `split` / channel shuffles, and the residual / bias additions, whose
gradients autograd routes.  fp32 on GPU tensors only; there is no PyTorch-op or CPU fallback.

Built for every family that has reference code: EdgeNeXt ('en') and MobileViT ('mv') backbones, Ghost-Dual-FPN ('gdf') and CSP-Dual-FPN
('cdf') necks, PointNet ('pn'), nano head.  PointNet++ (our own specification, no reference) raises NotImplementedError in training mode.
"""
import math

import torch

from . import train_functional as TF
from .train_ops import _SharedMLP1dFn, _LinearFn, _BmmPointsFn, _MaxPointsFn, _LogSoftmaxPointsFn

EDGENEXT = {  # edgenext_modules/model.py:14-66: depths, widths, heads, SDTA scales, depthwise kernel per stage
    'S0': dict(depths=[2, 2, 6, 2], heads=4, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
    'S1': dict(depths=[3, 3, 9, 3], heads=4, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
    'S2': dict(depths=[3, 3, 9, 3], heads=8, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
}
WIDTHS = {'S0': [32, 48, 96, 176], 'S1': [32, 48, 120, 224], 'S2': [32, 64, 144, 288]}        # neck/ghostdualfpn.py:20-25
BN_MOMENTUM = 0.1


MOBILEVIT = {  # backbone/vision/mobilevit_modules/mobilevit.py:225-240: channel plan (transformer widths are read from the weights)
    'S0': [16, 16, 32, 32, 48, 48, 96, 96, 96, 96, 176],
    'S1': [16, 32, 32, 32, 48, 48, 120, 120, 120, 120, 224],
    'S2': [16, 32, 32, 32, 64, 64, 144, 144, 144, 144, 288],
}


_POS_TABLES = {}        # (H, W, device, hidden, temperature) -> the Fourier position table [1, 64, H, W]


class TrainGraph:
    def __init__(self, model):
        self.m = model
        self.p = dict(model.named_parameters())
        self.b = dict(model.named_buffers())
        self.cfg = EDGENEXT[model.phi]
        self.w = WIDTHS[model.phi]
        self._nbt = []                    # the num_batches_tracked buffers of the BatchNorm layers this forward went through: incremented together at its end (one multi-tensor launch, not 80)

    # ---------------------------------------------------------------------------------------------- parameter access, layers
    def P(self, key):
        return self.p[key]

    def has(self, key):
        return key in self.p

    def bn(self, x, pfx, eps, relu=False):
        """nn.BatchNorm in training mode: batch statistics, running estimates and num_batches_tracked updated.  The BaseConv / SPP Conv
        layers are built with eps 1e-3 AND momentum 0.03 (normal_conv.py:45, spp.py:31); every other BatchNorm has torch's defaults."""
        self._nbt.append(self.b[pfx + '.num_batches_tracked'])
        return TF.batchnorm(x, self.P(pfx + '.weight'), self.P(pfx + '.bias'), self.b[pfx + '.running_mean'], self.b[pfx + '.running_var'],
                            True, 0.03 if eps == 1e-3 else BN_MOMENTUM, eps, relu)

    def conv(self, x, pfx, stride=1, pad=0, depthwise=False):
        w = self.P(pfx + '.weight')
        b = self.P(pfx + '.bias') if self.has(pfx + '.bias') else None
        if depthwise:
            return TF.dwconv(x, w, b)
        if w.shape[2] == 1 and w.shape[3] == 1 and stride == 1 and pad == 0:
            return TF.conv1x1(x, w, b)
        return TF.conv2d(x, w, b, stride, pad)

    def linear(self, x, pfx):
        """nn.Linear over the channels of a channels-first tensor [B, C, ...]."""
        return TF.conv1x1(x, self.P(pfx + '.weight'), self.P(pfx + '.bias') if self.has(pfx + '.bias') else None)

    def ln(self, x, pfx):
        """LayerNorm over the channels, eps 1e-6 (edgenext_modules/layers.py:5-26, both data formats)."""
        return TF.layernorm_channels(x, self.P(pfx + '.weight'), self.P(pfx + '.bias'), 1e-6)

    def base_conv(self, x, pfx, act=TF.ACT_RELU):
        """BaseConv 1x1: conv (no bias) + BN (eps 1e-3) + activation (backbone/conv_utils/normal_conv.py:36-52)."""
        y = self.bn(self.conv(x, pfx + '.conv'), pfx + '.bn', 1e-3, relu=act == TF.ACT_RELU)
        return y if act == TF.ACT_RELU else TF.act(y, act)

    def base_dwconv(self, x, pfx):
        """BaseConv with ds_conv: depthwise k x k -> pointwise 1x1 -> BN (1e-3) -> ReLU (normal_conv.py:23-33, 48-49)."""
        y = self.conv(x, pfx + '.conv.dconv', depthwise=True)
        return self.bn(self.conv(y, pfx + '.conv.pconv'), pfx + '.bn', 1e-3, relu=True)

    def ghost(self, x, pfx, oup, relu=True):
        """GhostModule (backbone/conv_utils/ghost_conv.py:6-29)."""
        x1 = self.bn(self.conv(x, pfx + '.primary_conv.0'), pfx + '.primary_conv.1', 1e-5, relu)
        x2 = self.bn(self.conv(x1, pfx + '.cheap_operation.0', depthwise=True), pfx + '.cheap_operation.1', 1e-5, relu)
        return torch.cat([x1, x2], 1)[:, :oup]

    def ghost_bottleneck(self, x, pfx, out_chs):
        """GhostBottleneck, stride 1, in != out (ghost_conv.py:32-70)."""
        y = self.ghost(self.ghost(x, pfx + '.ghost1', x.shape[1], True), pfx + '.ghost2', out_chs, False)
        s = self.bn(self.conv(x, pfx + '.shortcut.0', depthwise=True), pfx + '.shortcut.1', 1e-5)
        return y + self.bn(self.conv(s, pfx + '.shortcut.2'), pfx + '.shortcut.3', 1e-5)

    def upsample(self, x, pfx):
        """Upsample: BaseConv 1x1 + bilinear x2, align_corners (neck/ghostdualfpn.py:28-39)."""
        return TF.upsample2x(self.base_conv(x, pfx + '.upsample.0'))

    def shuffle_attention(self, x, pfx, G=4):
        """ShuffleAttention (backbone/attention_modules/shuffle_attention.py:48-72)."""
        b, c, h, w = x.shape
        x = x.reshape(b * G, c // G, h, w)
        c2 = c // (2 * G)
        x0, x1 = x[:, :c2].contiguous(), x[:, c2:].contiguous()
        gate = TF.channel_scale(TF.global_avg_pool(x0).view(b * G, c2, 1), self.P(pfx + '.cweight').reshape(c2)) + self.P(pfx + '.cbias').reshape(1, c2, 1)
        xc = TF.channel_scale(x0, TF.act(gate, TF.ACT_SIGMOID).view(b * G, c2))
        gn = TF.instance_norm(x1, self.P(pfx + '.gn.weight'), self.P(pfx + '.gn.bias'), 1e-5)
        gate = TF.channel_scale(gn, self.P(pfx + '.sweight').reshape(c2)) + self.P(pfx + '.sbias').reshape(1, c2, 1, 1)
        xs = TF.mul(x1, TF.act(gate, TF.ACT_SIGMOID))
        out = torch.cat([xc, xs], 1).reshape(b, c, h, w)
        return out.reshape(b, 2, c // 2, h, w).permute(0, 2, 1, 3, 4).reshape(b, c, h, w)          # channel_shuffle(2)

    def eca(self, x, pfx):
        """eca_block (backbone/attention_modules/eca.py:16-23): conv1d over the channel axis of the pooled vector."""
        wgt = self.P(pfx + '.conv.weight')                                   # [1, 1, k]
        k = wgt.shape[-1]
        B, C = x.shape[0], x.shape[1]
        m = TF.global_avg_pool(x).reshape(B, 1, C, 1)
        g = TF.conv2d(m, wgt.reshape(1, 1, k, 1), None, 1, ((k - 1) // 2, 0)).reshape(B, C)
        return TF.channel_scale(x, TF.act(g, TF.ACT_SIGMOID))

    # ---------------------------------------------------------------------------------------------- EdgeNeXt
    def pos_fourier(self, pfx, H, W, device, hidden=32, temperature=10000.0):
        """PositionalEncodingFourier (edgenext_modules/layers.py:38-59): an input-independent table, then its 1x1 projection (trainable)."""
        key = (H, W, str(device), hidden, temperature)
        if key in _POS_TABLES:                                                   # the table depends on (H, W) only: built and uploaded once (it used to be a host -> device copy per
            return self.conv(_POS_TABLES[key], pfx + '.token_projection')       # step — which also made the step impossible to capture in a graph)
        y = torch.arange(1, H + 1, dtype=torch.float32).view(H, 1).expand(H, W) / (float(H) + 1e-6) * (2 * math.pi)
        x = torch.arange(1, W + 1, dtype=torch.float32).view(1, W).expand(H, W) / (float(W) + 1e-6) * (2 * math.pi)
        dim_t = torch.arange(hidden, dtype=torch.float32)
        dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode='floor') / hidden)
        px, py = x[:, :, None] / dim_t, y[:, :, None] / dim_t
        px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
        py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
        pos = torch.cat((py, px), dim=2).permute(2, 0, 1).unsqueeze(0).contiguous().to(device)      # [1, 64, H, W]
        _POS_TABLES[key] = pos
        return self.conv(pos, pfx + '.token_projection')

    def mlp_tail(self, x, pfx):
        """norm -> pwconv1 -> GELU -> pwconv2 -> layer scale (conv_encoder.py:24-31, sdta_encoder.py:68-74), channels first."""
        y = self.ln(x, pfx + '.norm')
        y = TF.act(self.linear(y, pfx + '.pwconv1'), TF.ACT_GELU)
        return TF.channel_scale(self.linear(y, pfx + '.pwconv2'), self.P(pfx + '.gamma'))

    def conv_encoder(self, x, pfx):
        """ConvEncoder (edgenext_modules/conv_encoder.py:19-32)."""
        return x + self.mlp_tail(self.conv(x, pfx + '.dwconv', depthwise=True), pfx)

    def xca(self, t, pfx, heads):
        """XCA (edgenext_modules/sdta_encoder.py:162-185) on channels-first tokens t [B, C, N]: attention between CHANNELS."""
        B, C, N = t.shape
        d = C // heads
        qkv = self.linear(t, pfx + '.qkv').view(B, 3, heads, d, N)          # Linear's 3C outputs are ordered (which, head, channel)
        q = TF.l2_normalize_last(qkv[:, 0].reshape(B * heads, d, N))
        k = TF.l2_normalize_last(qkv[:, 1].reshape(B * heads, d, N))
        v = qkv[:, 2].reshape(B * heads, d, N)
        attn = TF.bmm_nt(q, k)                                               # [B*heads, d, d]
        attn = TF.row_scale(attn.view(B * heads, d * d), self.P(pfx + '.temperature').reshape(heads)).view(B * heads, d, d)
        o = TF.bmm_nn(TF.softmax_last(attn), v).view(B, C, N)
        return self.linear(o, pfx + '.proj')

    def sdta_encoder(self, x, pfx, scales, heads):
        """SDTAEncoder (edgenext_modules/sdta_encoder.py:39-74)."""
        B, C, H, W = x.shape
        width = max(int(math.ceil(C / scales)), int(math.floor(C // scales)))
        spx = torch.split(x, width, 1)
        outs, sp = [], None
        for i in range(scales - 1):
            sp = spx[i] if i == 0 else sp + spx[i]
            sp = self.conv(sp.contiguous(), f'{pfx}.convs.{i}', depthwise=True)
            outs.append(sp)
        t = torch.cat(outs + [spx[scales - 1]], 1)
        if self.has(pfx + '.pos_embd.token_projection.weight'):
            t = t + self.pos_fourier(pfx + '.pos_embd', H, W, x.device)
        t = t.reshape(B, C, H * W)
        t = t + TF.channel_scale(self.xca(self.ln(t, pfx + '.norm_xca'), pfx + '.xca', heads), self.P(pfx + '.gamma_xca'))
        return x + self.mlp_tail(t.view(B, C, H, W), pfx)

    def edgenext(self, x, pfx):
        """EdgeNeXt.forward_features (edgenext_modules/edgenext.py:73-86)."""
        feats = []
        for i in range(4):
            d = f'{pfx}.downsample_layers.{i}'
            if i == 0:
                x = self.ln(self.conv(x, d + '.0', stride=4), d + '.1')
            else:
                x = self.conv(self.ln(x, d + '.0'), d + '.1', stride=2)
            depth = self.cfg['depths'][i]
            for j in range(depth):
                blk = f'{pfx}.stages.{i}.{j}'
                if i > 0 and j == depth - 1:                                  # global_block = [0, 1, 1, 1]: the last block of stages 1-3
                    x = self.sdta_encoder(x, blk, self.cfg['scales'][i], self.cfg['heads'])
                else:
                    x = self.conv_encoder(x, blk)
            feats.append(x)
        return feats

    # ---------------------------------------------------------------------------------------------- MobileViT
    def conv_bn_silu(self, x, pfx, stride=1, pad=0):
        """conv (no bias) + BatchNorm (1e-5) + SiLU (mobilevit.py:8-21)."""
        return TF.act(self.bn(self.conv(x, pfx + '.0', stride=stride, pad=pad), pfx + '.1', 1e-5), TF.ACT_SILU)

    def mv2block(self, x, pfx, stride, oup):
        """MV2Block with expansion (mobilevit.py:93-131).  The stride-2 depthwise 3x3 (pad 1) is the stride-1 result at the even positions."""
        y = self.conv_bn_silu(x, pfx + '.conv')
        y = self.conv(y, pfx + '.conv.3', depthwise=True)
        if stride == 2:
            y = y[:, :, ::2, ::2].contiguous()
        y = TF.act(self.bn(y, pfx + '.conv.4', 1e-5), TF.ACT_SILU)
        y = self.bn(self.conv(y, pfx + '.conv.6'), pfx + '.conv.7', 1e-5)
        return x + y if (stride == 1 and x.shape[1] == oup) else y

    def ln5(self, t, pfx):
        """nn.LayerNorm(dim), eps 1e-5, over the channel axis of channels-first tokens [T, D, N]."""
        return TF.layernorm_channels(t, self.P(pfx + '.weight'), self.P(pfx + '.bias'), 1e-5)

    def mv_transformer(self, t, pfx, depth, heads=4, dim_head=8):
        """Transformer / Attention / FeedForward (mobilevit.py:33-90) on channels-first tokens t [T = B * patches, D, N]."""
        T, D, N = t.shape
        for l in range(depth):
            a = f'{pfx}.layers.{l}.0'
            qkv = TF.conv1x1(self.ln5(t, a + '.norm'), self.P(a + '.fn.to_qkv.weight')).view(T, 3, heads, dim_head, N)
            q, k, v = (qkv[:, i].reshape(T * heads, dim_head, N) for i in range(3))
            attn = TF.bmm_nt(q.transpose(1, 2).contiguous(), k.transpose(1, 2).contiguous())                   # [T*heads, N(query), N(key)]
            attn = TF.softmax_last(TF.row_scale(attn, torch.full((1,), dim_head ** -0.5, device=t.device)))
            o = TF.bmm_nt(v, attn).reshape(T, heads * dim_head, N)                                              # o[d, i] = sum_j v[d, j] attn[i, j]
            t = TF.conv1x1(o, self.P(a + '.fn.to_out.0.weight'), self.P(a + '.fn.to_out.0.bias')) + t
            f = f'{pfx}.layers.{l}.1'
            h = TF.act(self.linear(self.ln5(t, f + '.norm'), f + '.fn.net.0'), TF.ACT_SILU)
            t = self.linear(h, f + '.fn.net.3') + t
        return t

    def mvit_block(self, x, pfx, depth):
        """MobileViTBlock (mobilevit.py:147-165), 2x2 patches: 'b d (h ph) (w pw) -> b (ph pw) (h w) d' as channels-first token groups."""
        y = x
        x = self.conv_bn_silu(self.conv_bn_silu(x, pfx + '.conv1', 1, 1), pfx + '.conv2')
        B, D, H, W = x.shape
        h, w = H // 2, W // 2
        t = x.reshape(B, D, h, 2, w, 2).permute(0, 3, 5, 1, 2, 4).reshape(B * 4, D, h * w).contiguous()
        t = self.mv_transformer(t, pfx + '.transformer', depth)
        x = t.reshape(B, 2, 2, D, h, w).permute(0, 3, 4, 1, 5, 2).reshape(B, D, H, W).contiguous()
        x = self.conv_bn_silu(x, pfx + '.conv3')
        return self.conv_bn_silu(torch.cat((x, y), 1), pfx + '.conv4', 1, 1)

    def mobilevit(self, x, pfx):
        """MobileViT.forward (mobilevit.py:198-222)."""
        ch = MOBILEVIT[self.m.phi]
        x = self.conv_bn_silu(x, pfx + '.conv1', 2, 1)
        x = self.mv2block(x, pfx + '.mv2.0', 1, ch[1])
        x = self.mv2block(x, pfx + '.mv2.1', 2, ch[2])
        x = self.mv2block(x, pfx + '.mv2.2', 1, ch[3])
        f2 = x = self.mv2block(x, pfx + '.mv2.3', 1, ch[3])
        f3 = x = self.mvit_block(self.mv2block(x, pfx + '.mv2.4', 2, ch[4]), pfx + '.mvit.0', 2)
        f4 = x = self.mvit_block(self.mv2block(x, pfx + '.mv2.5', 2, ch[6]), pfx + '.mvit.1', 4)
        x = self.mvit_block(self.mv2block(x, pfx + '.mv2.6', 2, ch[8]), pfx + '.mvit.2', 3)
        return [f2, f3, f4, self.conv_bn_silu(x, pfx + '.conv2')]

    # ---------------------------------------------------------------------------------------------- CSP neck blocks
    def base_conv_k(self, x, pfx, act, k=1):
        """BaseConv: conv k x k (pad (k-1)//2, no bias) + BN (1e-3) + activation (normal_conv.py:36-47)."""
        y = self.bn(self.conv(x, pfx + '.conv', pad=(k - 1) // 2), pfx + '.bn', 1e-3, relu=act == TF.ACT_RELU)
        return y if act == TF.ACT_RELU else TF.act(y, act)

    def csp_bottleneck(self, x, pfx, cout):
        """Bottleneck (neck/cspdualfpn.py:42-57): 1x1 (SiLU) -> 3x3 (ReLU), + x when in == out."""
        y = self.base_conv_k(self.base_conv_k(x, pfx + '.conv1', TF.ACT_SILU), pfx + '.conv2', TF.ACT_RELU, 3)
        return y + x if x.shape[1] == cout else y

    def csp_layer(self, x, pfx):
        """CSPLayer, n = 1 (neck/cspdualfpn.py:60-78)."""
        x1 = self.base_conv_k(x, pfx + '.conv1', TF.ACT_SILU)
        x2 = self.base_conv_k(x, pfx + '.conv2', TF.ACT_SILU)
        x1 = self.csp_bottleneck(x1, pfx + '.m.0', x1.shape[1])
        return self.base_conv_k(torch.cat((x1, x2), 1), pfx + '.conv3', TF.ACT_SILU)

    # ---------------------------------------------------------------------------------------------- neck, decoders
    def spp(self, x, pfx):
        """SPP / SPPF (neck/spp.py:41-67): Conv = conv + BN (1e-3) + SiLU."""
        y = self.base_conv(x, pfx + '.cv1', TF.ACT_SILU)
        if self.m.spp:
            pools = [TF.maxpool_same(y, k) for k in (5, 9, 13)]
        else:
            y1 = TF.maxpool_same(y, 5)
            y2 = TF.maxpool_same(y1, 5)
            pools = [y1, y2, TF.maxpool_same(y2, 5)]
        return self.base_conv(torch.cat([y] + pools, 1), pfx + '.cv2', TF.ACT_SILU)

    def ghost_dual_fpn(self, x):
        """GhostDualFPN.forward (neck/ghostdualfpn.py:156-200)."""
        f = 'image_radar_encoder.fpn'
        w = self.w
        m2, m3, m4, m5 = self.edgenext(x, f + '.backbone') if self.m.backbone == 'en' else self.mobilevit(x, f + '.backbone')
        csp = self.m.neck == 'cdf'                       # CSPDualFPN.forward (neck/cspdualfpn.py:193-237): the same graph with CSP blocks
        p5 = self.spp(m5, f + '.spp')
        c4 = torch.cat([self.upsample(p5, f + '.upsample_5_to_4'), m4], 1)
        p4 = self.csp_layer(c4, f + '.ghost_5_to_4') if csp else self.ghost_bottleneck(c4, f + '.ghost_5_to_4', w[2])
        c3 = torch.cat([self.upsample(p4, f + '.upsample_4_to_3'), m3], 1)
        p3 = self.csp_layer(c3, f + '.ghost_4_to_3') if csp else self.ghost_bottleneck(c3, f + '.ghost_4_to_3', w[1])
        outs = {}
        for name, sa, oup in (('lane', 'stage_3_lane_seg', 2), ('se', 'stage_3_semantic_seg', self.m.num_seg)):
            y = self.shuffle_attention(p3, f'{f}.{sa}')
            for lvl, c in (('3_to_2', w[1]), ('2_to_1', w[0]), ('1_to_0', w[0])):
                y = self.upsample(y, f'{f}.{name}_seg_{lvl}')
                y = self.csp_bottleneck(y, f'{f}.{name}_seg_ghost_{lvl}', c) if csp else self.ghost(y, f'{f}.{name}_seg_ghost_{lvl}', c)
            outs[name] = self.csp_bottleneck(y, f'{f}.{name}_seg_head', oup) if csp else self.ghost(y, f'{f}.{name}_seg_head', oup)
        return outs['se'], outs['lane'], (p5 + m5, p4 + m4, p3 + m3)

    # ---------------------------------------------------------------------------------------------- radar branch, fusion, head
    def rc_block(self, x, pfx, down):
        """RCBlock / RadarConv / DeformableConv2d (backbone/radar/RadarEncoder.py:38-74, conv_utils/dcn.py:49-63)."""
        d = pfx + '.radar_conv.deformable_conv'
        y = TF.avgpool3(x)
        off = self.conv(y, d + '.offset_conv', pad=1)
        msk = TF.row_scale(TF.act(self.conv(y, d + '.modulator_conv', pad=1), TF.ACT_SIGMOID), torch.full((1,), 2.0, device=x.device))
        y = TF.deform_conv3x3(y, off, msk, self.P(d + '.regular_conv.weight'), 1, 1)
        y = x + self.bn(self.conv(y, pfx + '.weight_conv1'), pfx + '.norm', 1e-5, relu=True)
        return self.conv(y, pfx + '.weight_conv2', stride=2, pad=1) if down else self.conv(y, pfx + '.weight_conv2')

    def rcnet(self, x):
        """RCNet.forward (RadarEncoder.py:99-109)."""
        down = [True, True, False, True, False, True, False, True]
        outs = []
        for i in range(8):
            x = self.rc_block(x, f'image_radar_encoder.radar_encoder.rc_blocks.{i}', down[i])
            if i > 1 and i % 2 == 1:
                outs.append(x)
        return outs

    def fuse(self, img, rad, stage):
        """IREncoder fusion (backbone/IREncoder.py:79-89)."""
        e = 'image_radar_encoder'
        z = torch.cat([self.eca(img, f'{e}.channel_attn_stage{stage}.0'), self.eca(rad, f'{e}.channel_attn_stage{stage}.1')], 1)
        return self.bn(z, f'{e}.norm_stage{stage}', 1e-5, relu=True)

    def head(self, feats):
        """DecoupleHead.forward (head/decouplehead.py:58-103); the width (64 nano / 256) is the weights'."""
        outs = []
        for k, x in enumerate(feats):
            x = self.base_conv(x, f'det_head.stems.{k}')
            c = self.base_dwconv(self.base_dwconv(x, f'det_head.cls_convs.{k}.0'), f'det_head.cls_convs.{k}.1')
            r = self.base_dwconv(self.base_dwconv(x, f'det_head.reg_convs.{k}.0'), f'det_head.reg_convs.{k}.1')
            outs.append(torch.cat([self.conv(r, f'det_head.reg_preds.{k}'), self.conv(r, f'det_head.obj_preds.{k}'), self.conv(c, f'det_head.cls_preds.{k}')], 1))
        return outs

    # ---------------------------------------------------------------------------------------------- PointNet
    def shared_mlp(self, x, conv, bn, relu=True):
        """Conv1d(k=1) / Linear + BatchNorm1d [+ ReLU] on [B, C, N] (pointnet_utils.py:29-31, 69-71, 124-127)."""
        self._nbt.append(self.b[bn + '.num_batches_tracked'])
        return _SharedMLP1dFn.apply(x, self.P(conv + '.weight'), self.P(conv + '.bias'), self.P(bn + '.weight'), self.P(bn + '.bias'),
                                    self.b[bn + '.running_mean'], self.b[bn + '.running_var'], True, BN_MOMENTUM, 1e-5, relu)

    def fc_bn(self, x, fc, bn):
        """Linear + BatchNorm1d + ReLU on [B, C]: the batch is the axis of the statistics, i.e. the shared MLP on [1, C, B]."""
        return self.shared_mlp(x.t().contiguous().unsqueeze(0), fc, bn).squeeze(0).t()

    def stn(self, x, pfx, k):
        """STN3d / STNkd (pointnet_utils.py:10-85)."""
        y = self.shared_mlp(x, pfx + '.conv1', pfx + '.bn1')
        y = self.shared_mlp(y, pfx + '.conv2', pfx + '.bn2')
        y = _MaxPointsFn.apply(self.shared_mlp(y, pfx + '.conv3', pfx + '.bn3'))
        y = self.fc_bn(self.fc_bn(y, pfx + '.fc1', pfx + '.bn4'), pfx + '.fc2', pfx + '.bn5')
        y = _LinearFn.apply(y.t().contiguous().unsqueeze(0), self.P(pfx + '.fc3.weight'), self.P(pfx + '.fc3.bias')).squeeze(0).t()
        return (y + torch.eye(k, dtype=y.dtype, device=y.device).reshape(1, k * k)).view(-1, k, k)

    def pointnet(self, pts):
        """PointNet_SEG.forward / PointNetEncoder.forward (pointnet_sem_seg.py:26-37, pointnet_utils.py:103-133)."""
        p = 'pc_seg_model'
        B, D, N = pts.shape
        trans = self.stn(pts, p + '.feat.stn', 3)
        xyz = _BmmPointsFn.apply(pts[:, :3].contiguous(), trans)
        x = torch.cat([xyz, pts[:, 3:]], 1) if D > 3 else xyz
        x = self.shared_mlp(x, p + '.feat.conv1', p + '.feat.bn1')
        x = _BmmPointsFn.apply(x, self.stn(x, p + '.feat.fstn', 32))
        pointfeat = x
        x = self.shared_mlp(x, p + '.feat.conv2', p + '.feat.bn2')
        g = _MaxPointsFn.apply(self.shared_mlp(x, p + '.feat.conv3', p + '.feat.bn3', relu=False))
        x = torch.cat([g.unsqueeze(2).expand(-1, -1, N), pointfeat], 1)
        for i in (1, 2, 3):
            x = self.shared_mlp(x, f'{p}.conv{i}', f'{p}.bn{i}')
        return _LogSoftmaxPointsFn.apply(_LinearFn.apply(x, self.P(p + '.conv4.weight'), self.P(p + '.conv4.bias')))

    # ---------------------------------------------------------------------------------------------- PointNet++ (OUR OWN specification, DESIGN 5b / spec.py::PN2)
    def rows_mlp(self, rows, conv, bn):
        """Conv (k = 1) + BatchNorm + ReLU over ROWS [R, C]: the statistics run over all rows — BatchNorm2d over (B, centroids, samples) of a set-abstraction level,
        BatchNorm1d over (B, points) of a propagation level — i.e. the shared MLP on [1, C, R]."""
        return self.shared_mlp(rows.t().contiguous().unsqueeze(0), conv, bn).squeeze(0).t().contiguous()

    def pointnet2(self, pts):
        """Set abstraction x 4 (farthest-point sampling, ball query, shared MLP, max over the ball), feature propagation x 4 (3-NN interpolation + skip + shared MLP), head.
        The reference snapshot has no PointNet++ code (nets/Achelous.py:31-32): structure, widths, radii and tie rules are spec.py::PN2's, the same ones the inference
        engine and oracle/pointnet2_oracle.py follow; the geometry kernels ARE the inference engine's (k_pn2.h at fp32), so a training forward selects exactly the
        points an inference forward selects."""
        import numpy as np
        from .spec import PN2_VARIANTS, pn2_scales
        PN2 = PN2_VARIANTS[self.m.pc_seg_kind]
        p = 'pc_seg_model'
        B, D, N = pts.shape
        rows = pts.transpose(1, 2).contiguous()                                   # [B, N, D]
        levels = [(rows[:, :, :3].contiguous(), rows)]
        for k, cfg in enumerate(PN2['sa']):
            xyz, feats = levels[-1]
            S = N // cfg['div']
            _, new_xyz = TF.pn2_fps(xyz, S)
            outs = []
            for j, sc in enumerate(pn2_scales(cfg)):                              # one (radius, nsample, MLP) stack — or two, on the same centroids (pn2_msg)
                K = sc['nsample']
                h = TF.pn2_group(xyz, new_xyz, feats, K, float(np.float32(sc['radius'] * sc['radius'])))        # [B*S*K, 3 + C]
                x = h.t().contiguous().unsqueeze(0)
                for i in range(len(sc['mlp'])):
                    names = (f'{p}.sa{k + 1}.conv_blocks.{j}.{i}', f'{p}.sa{k + 1}.bn_blocks.{j}.{i}') if 'scales' in cfg else (f'{p}.sa{k + 1}.mlp_convs.{i}', f'{p}.sa{k + 1}.mlp_bns.{i}')
                    x = self.shared_mlp(x, *names)
                cout = x.shape[1]
                f = _MaxPointsFn.apply(x.view(cout, B * S, K))                   # max over the ball: [cout, B*S]
                outs.append(f.t().contiguous().view(B, S, cout))
            levels.append((new_xyz, outs[0] if len(outs) == 1 else torch.cat(outs, 2).contiguous()))
        cur = levels[-1][1]
        L = len(PN2['sa'])
        for j, widths in enumerate(PN2['fp']):                                   # fp4 .. fp1
            lvl = L - 1 - j
            xyz1, f1 = levels[lvl]
            h = TF.pn2_interp(xyz1, levels[lvl + 1][0], f1 if lvl > 0 else None, cur)                           # [B*n, C1 + C2]
            for i in range(len(widths)):
                h = self.rows_mlp(h, f'{p}.fp{lvl + 1}.mlp_convs.{i}', f'{p}.fp{lvl + 1}.mlp_bns.{i}')
            cur = h.view(B, xyz1.shape[1], widths[-1])
        h = self.rows_mlp(cur.reshape(B * N, -1), p + '.conv1', p + '.bn1')
        z = _LinearFn.apply(h.view(B, N, -1).transpose(1, 2).contiguous(), self.P(p + '.conv2.weight'), self.P(p + '.conv2.bias'))       # [B, classes, N]
        return _LogSoftmaxPointsFn.apply(z)

    # ---------------------------------------------------------------------------------------------- Achelous.forward (nets/Achelous.py:49-53)
    def forward(self, x, x_radar, x_pc):
        for t in (x, x_radar, x_pc):
            if t is not None and t.dtype != torch.float32:
                raise TypeError("training mode runs in float32")
        pc = None                                                                    # Achelous3T: no point stream
        if x_pc is not None:
            pc = self.pointnet2(x_pc.contiguous()) if self.m.pc_seg_kind in ('pn2', 'pn2_msg') else self.pointnet(x_pc.contiguous())
        se, lane, (q5, q4, q3) = self.ghost_dual_fpn(x.contiguous())
        r3, r4, r5 = self.rcnet(x_radar.contiguous())
        det = self.head((self.fuse(q3, r3, 3), self.fuse(q4, r4, 4), self.fuse(q5, r5, 5)))
        self.flush_counters()
        return det, se, lane, pc

    def flush_counters(self):
        """num_batches_tracked += 1 for every BatchNorm layer evaluated since the last flush (callers that evaluate a branch on its own — the tests — call this themselves)."""
        if self._nbt:
            with torch.no_grad():
                torch._foreach_add_(self._nbt, 1)
            self._nbt = []


class GraphedTrainStep:
    """One whole training step — `model(x, radar, points)` in `.train()`, `loss_fn`, `loss.backward()`, `optimizer.step()` — captured ONCE into a HIP graph and replayed
    (round 6).  A step of this network is ~1 300 small launches issued from Python through ctypes; below batch ~16 the host cannot issue them as fast as the
    GPU retires them (batch 8: 31 ms eager, of which the GPU needs 22).  A replayed graph has no host side: batch 8 21.8 ms (-30 %), batch 16 30.2 ms (-3 %), batch 32 unchanged
    (47 ms: GPU-bound).  The reference's own loop (utils/utils_fit.py:37-166) is eager and keeps working unchanged; this is the opt-in for small-batch fine-tuning:

        step = GraphedTrainStep(net, opt, loss_fn, (images, radar, points), (targets...))     # captures; the model / optimizer state is left exactly as it was
        for images, radar, points, *targets in loader:
            loss = step(images, radar, points, *targets)                                      # copies into the captured buffers, replays; `loss` is a device tensor

    Requirements (those of any whole-step capture): fixed shapes and dtypes (the example tensors'), a `loss_fn(outputs, *targets) -> scalar tensor` made of torch ops without host
    synchronisation (no `.item()`, no data-dependent Python control flow), an optimizer whose step is capture-safe (SGD; Adam / AdamW with `capturable=True`), and no change of
    `requires_grad` flags or parameter identity afterwards.  `outputs` = `(det_list, se, lane, pc)` as `forward` returns them; `step.outputs` holds the last replay's.
    The warm-up steps that precede the capture run on the example batch and are UNDONE: parameters, BatchNorm statistics and optimizer state are restored in place.
    If the model has already run EAGER training steps, drop every reference to their outputs / losses first: a live autograd graph keeps the parameters' AccumulateGrad nodes
    bound to the eager steps' stream, and replaying them inside a capture is not legal (measured: hipStreamEndCapture crashes)."""

    def __init__(self, model, optimizer, loss_fn, example_inputs, example_targets=(), warmup=3):
        if not model.training:
            raise RuntimeError("GraphedTrainStep captures a TRAINING step: call model.train() first")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.inputs = tuple(t.detach().clone() if t is not None else None for t in example_inputs)
        self.targets = tuple(t.detach().clone() for t in example_targets)
        dev = self.inputs[0].device
        if dev.type != 'cuda':
            raise RuntimeError("GraphedTrainStep needs GPU tensors")
        import gc
        gc.collect()                                                          # (unreferenced autograd graphs of earlier eager steps release their AccumulateGrad nodes)
        saved_model = {k: v.detach().clone() for k, v in model.state_dict().items()}
        had_state = len(optimizer.state) > 0
        saved_opt = {id(p): {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in st.items()} for p, st in optimizer.state.items()} if had_state else None

        def run():
            outs = model(*self.inputs)
            loss = loss_fn(outs, *self.targets)
            loss.backward()
            optimizer.step()
            return outs, loss
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, int(warmup))):                              # allocator pools, workspaces, lazily built optimizer state: everything a capture may not create
                optimizer.zero_grad(set_to_none=True)
                run()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.outputs, self.loss = run()
        torch.cuda.synchronize(dev)
        # undo the warm-up steps (a capture itself executes nothing): parameters and buffers back in place; optimizer state back to what it was — or, if it had none, to zeros,
        # which is what the first real step of a fresh optimizer computes from (SGD: buf = grad; Adam: moments and step count from zero)
        with torch.no_grad():
            for k, v in model.state_dict().items():
                v.copy_(saved_model[k])
            for p, st in optimizer.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):
                        if had_state and id(p) in saved_opt and k in saved_opt[id(p)]:
                            v.copy_(saved_opt[id(p)][k])
                        else:
                            v.zero_()
        self.steps = 0

    @torch.no_grad()
    def _load(self, inputs, targets):
        if len(inputs) != len(self.inputs) or len(targets) != len(self.targets):
            raise ValueError("GraphedTrainStep: the number of inputs / targets differs from the captured step's")
        for dst, src in zip(self.inputs + self.targets, tuple(inputs) + tuple(targets)):
            if dst is None:
                continue
            if src.shape != dst.shape or src.dtype != dst.dtype:
                raise ValueError(f"GraphedTrainStep: captured {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype} (a graph has fixed shapes: build another step)")
            dst.copy_(src, non_blocking=True)

    def __call__(self, x, x_radar, x_points, *targets):
        self._load((x, x_radar, x_points), targets)
        self.graph.replay()
        self.steps += 1
        return self.loss.detach()
