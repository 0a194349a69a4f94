"""CPU fp32 restatement of `Achelous.forward` + `decode_outputs` + `non_max_suppression`.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): imported by tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() as the checker; never by achelous_amd/.

Written from scratch as a *functional* evaluator over a reference-keyed state_dict (no nn.Module
tree): every function cites the reference file:line whose arithmetic it restates.  Pinned against the
imported reference by tests/golden/gen_golden.py (which asserts oracle == reference before writing any
fixture) and, on the GPU box where /root/reference does not exist, against the committed fixtures by
tests/test_oracle_golden.py.

Inference (eval-mode) semantics only: BatchNorm uses running statistics, DropPath/Dropout are identities
(SURVEY.md §9).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from .deform_conv import deform_conv2d
from .nms import batched_nms_np
from .pointnet2_oracle import PN2, PN2_MSG, PointNet2Oracle

WIDTHS = {'S0': [32, 48, 96, 176], 'S1': [32, 48, 120, 224], 'S2': [32, 64, 144, 288]}  # neck/ghostdualfpn.py:20-25

EDGENEXT = {  # backbone/vision/edgenext_modules/model.py:14-66
    'S0': dict(depths=[2, 2, 6, 2], dims=[32, 48, 96, 176], heads=4, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
    'S1': dict(depths=[3, 3, 9, 3], dims=[32, 48, 120, 224], heads=4, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
    'S2': dict(depths=[3, 3, 9, 3], dims=[32, 64, 144, 288], heads=8, scales=[2, 2, 3, 4], ks=[3, 5, 7, 9]),
}
MOBILEVIT = {  # backbone/vision/mobilevit_modules/mobilevit.py:225-240
    'S0': dict(dims=[64, 80, 96], ch=[16, 16, 32, 32, 48, 48, 96, 96, 96, 96, 176], exp=2),
    'S1': dict(dims=[96, 120, 144], ch=[16, 32, 32, 32, 48, 48, 120, 120, 120, 120, 224], exp=4),
    'S2': dict(dims=[144, 192, 240], ch=[16, 32, 32, 32, 64, 64, 144, 144, 144, 144, 288], exp=4),
}


class AchelousOracle:
    def __init__(self, state_dict, num_det=7, num_seg=9, phi='S0', backbone='en', neck='gdf', pc_seg='pn',
                 pc_channels=5, pc_classes=8, nano_head=True, spp=True, resolution=320, boundary_dtype=None):
        if neck not in ('gdf', 'cdf') or backbone not in ('en', 'mv') or pc_seg not in ('pn', 'pn2', 'pn2_msg'):
            raise NotImplementedError("oracle covers backbone in {en,mv}, neck in {gdf,cdf}, pc_seg in {pn,pn2,pn2_msg}")
        self.neck = neck
        # 'pn2' has no reference implementation: own specification, self-oracle, parity unpinned (pointnet2_oracle.py)
        self.pn2 = PointNet2Oracle(state_dict, {'pn2': PN2, 'pn2_msg': PN2_MSG}[pc_seg]) if pc_seg in ('pn2', 'pn2_msg') else None
        self.sd = {k: (v.detach().to(device='cpu', dtype=torch.float32) if v.is_floating_point() else v.detach().cpu())
                   for k, v in state_dict.items()}
        self.num_det, self.num_seg, self.phi, self.backbone = num_det, num_seg, phi, backbone
        self.pc_channels, self.pc_classes, self.nano_head, self.spp = pc_channels, pc_classes, nano_head, spp
        self.w = WIDTHS[phi]
        self.taps = {}
        # boundary_dtype=torch.bfloat16: the "ideal bf16-storage engine" — fp32 arithmetic everywhere, but the image / radar inputs and every
        # SURVEY 8(a) boundary tensor of the image and radar paths (the tensors the taps name) are ROUNDED to that type before they are
        # passed on.  It is the yardstick for what bf16 storage alone costs on a given set of weights and frames (tests/test_gpu_parity.py).
        self.boundary_dtype = boundary_dtype

    def _b(self, name, t):
        """Record a boundary tensor; with boundary_dtype set, hand on its rounded value."""
        if self.boundary_dtype is not None:
            t = t.to(self.boundary_dtype).float()
        self.taps[name] = t
        return t

    # ------------------------------------------------------------------ primitives
    def P(self, key):
        return self.sd[key]

    def has(self, key):
        return key in self.sd

    def bn(self, x, pfx, eps):
        """Eval-mode BatchNorm (running statistics)."""
        shape = [1, -1] + [1] * (x.dim() - 2)
        m, v = self.P(pfx + '.running_mean').view(shape), self.P(pfx + '.running_var').view(shape)
        g, b = self.P(pfx + '.weight').view(shape), self.P(pfx + '.bias').view(shape)
        return (x - m) / torch.sqrt(v + eps) * g + b

    def conv(self, x, pfx, stride=1, pad=0, groups=1):
        b = self.P(pfx + '.bias') if self.has(pfx + '.bias') else None
        return F.conv2d(x, self.P(pfx + '.weight'), b, stride=stride, padding=pad, groups=groups)

    def base_conv(self, x, pfx, act='relu'):
        """BaseConv 1x1: conv(no bias) + BN(eps 1e-3) + act  (backbone/conv_utils/normal_conv.py:36-52)."""
        y = self.bn(self.conv(x, pfx + '.conv'), pfx + '.bn', 1e-3)
        return F.relu(y) if act == 'relu' else y * torch.sigmoid(y)

    def base_dwconv(self, x, pfx, k=5):
        """BaseConv with ds_conv=True: dw kxk -> pw 1x1 (no BN/act in between) -> BN(1e-3) -> ReLU
        (normal_conv.py:23-33,48-49)."""
        c = x.shape[1]
        y = F.conv2d(x, self.P(pfx + '.conv.dconv.weight'), None, padding=(k - 1) // 2, groups=c)
        y = F.conv2d(y, self.P(pfx + '.conv.pconv.weight'), None)
        return F.relu(self.bn(y, pfx + '.bn', 1e-3))

    def ghost(self, x, pfx, oup, relu=True):
        """GhostModule (backbone/conv_utils/ghost_conv.py:6-29), BN eps 1e-5."""
        x1 = self.bn(self.conv(x, pfx + '.primary_conv.0'), pfx + '.primary_conv.1', 1e-5)
        if relu:
            x1 = F.relu(x1)
        init = x1.shape[1]
        x2 = self.bn(self.conv(x1, pfx + '.cheap_operation.0', pad=1, groups=init), pfx + '.cheap_operation.1', 1e-5)
        if relu:
            x2 = F.relu(x2)
        return torch.cat([x1, x2], 1)[:, :oup]

    def ghost_bottleneck(self, x, pfx, out_chs):
        """GhostBottleneck stride 1, in != out (ghost_conv.py:32-70)."""
        mid = x.shape[1]
        y = self.ghost(x, pfx + '.ghost1', mid, relu=True)
        y = self.ghost(y, pfx + '.ghost2', out_chs, relu=False)
        s = self.bn(self.conv(x, pfx + '.shortcut.0', pad=1, groups=x.shape[1]), pfx + '.shortcut.1', 1e-5)
        s = self.bn(self.conv(s, pfx + '.shortcut.2'), pfx + '.shortcut.3', 1e-5)
        return y + s

    def upsample(self, x, pfx):
        """Upsample = BaseConv 1x1 + bilinear x2 align_corners=True (neck/ghostdualfpn.py:28-39)."""
        y = self.base_conv(x, pfx + '.upsample.0')
        return F.interpolate(y, scale_factor=2, mode='bilinear', align_corners=True)

    def shuffle_attention(self, x, pfx, G=4):
        """ShuffleAttention (backbone/attention_modules/shuffle_attention.py:48-72)."""
        b, c, h, w = x.shape
        x = x.reshape(b * G, c // G, h, w)
        x0, x1 = x[:, :c // (2 * G)], x[:, c // (2 * G):]
        xc = x0 * torch.sigmoid(self.P(pfx + '.cweight') * x0.mean((2, 3), keepdim=True) + self.P(pfx + '.cbias'))
        mu = x1.mean((2, 3), keepdim=True)
        var = x1.var((2, 3), unbiased=False, keepdim=True)
        gn = (x1 - mu) / torch.sqrt(var + 1e-5) * self.P(pfx + '.gn.weight').view(1, -1, 1, 1) \
            + self.P(pfx + '.gn.bias').view(1, -1, 1, 1)
        xs = x1 * torch.sigmoid(self.P(pfx + '.sweight') * gn + self.P(pfx + '.sbias'))
        out = torch.cat([xc, xs], 1).reshape(b, c, h, w)
        return out.reshape(b, 2, c // 2, h, w).permute(0, 2, 1, 3, 4).reshape(b, c, h, w)  # channel_shuffle(2)

    def eca(self, x, pfx):
        """eca_block (backbone/attention_modules/eca.py:16-23)."""
        wgt = self.P(pfx + '.conv.weight')
        k = wgt.shape[-1]
        m = x.mean((2, 3))                                   # [B, C]
        g = F.conv1d(m.unsqueeze(1), wgt, padding=(k - 1) // 2).squeeze(1)
        return x * torch.sigmoid(g)[:, :, None, None]

    @staticmethod
    def ln_last(x, w, b, eps):
        return F.layer_norm(x, (x.shape[-1],), w, b, eps)

    # ------------------------------------------------------------------ EdgeNeXt (a2-a5)
    def pos_fourier(self, pfx, H, W, hidden=32, temperature=10000.0):
        """PositionalEncodingFourier (edgenext_modules/layers.py:38-59); input independent."""
        dev = self.P(pfx + '.token_projection.weight').device           # (CPU everywhere except profiles/scripts/train_step.py --baseline)
        y = torch.arange(1, H + 1, dtype=torch.float32, device=dev).view(H, 1).expand(H, W)
        x = torch.arange(1, W + 1, dtype=torch.float32, device=dev).view(1, W).expand(H, W)
        y = y / (float(H) + 1e-6) * (2 * math.pi)
        x = x / (float(W) + 1e-6) * (2 * math.pi)
        dim_t = torch.arange(hidden, dtype=torch.float32, device=dev)
        dim_t = temperature ** (2 * torch.div(dim_t, 2, rounding_mode='floor') / hidden)
        px = x[:, :, None] / dim_t
        py = y[:, :, None] / dim_t
        px = torch.stack((px[:, :, 0::2].sin(), px[:, :, 1::2].cos()), dim=3).flatten(2)
        py = torch.stack((py[:, :, 0::2].sin(), py[:, :, 1::2].cos()), dim=3).flatten(2)
        pos = torch.cat((py, px), dim=2).permute(2, 0, 1).unsqueeze(0)         # [1, 64, H, W]
        return self.conv(pos, pfx + '.token_projection')                           # [1, C, H, W]

    def conv_encoder(self, x, pfx, k):
        """ConvEncoder (edgenext_modules/conv_encoder.py:19-32)."""
        c = x.shape[1]
        y = self.conv(x, pfx + '.dwconv', pad=k // 2, groups=c).permute(0, 2, 3, 1)
        y = self.ln_last(y, self.P(pfx + '.norm.weight'), self.P(pfx + '.norm.bias'), 1e-6)
        y = F.linear(y, self.P(pfx + '.pwconv1.weight'), self.P(pfx + '.pwconv1.bias'))
        y = F.gelu(y)
        y = F.linear(y, self.P(pfx + '.pwconv2.weight'), self.P(pfx + '.pwconv2.bias'))
        y = self.P(pfx + '.gamma') * y
        return x + y.permute(0, 3, 1, 2)

    def xca(self, t, pfx, heads):
        """XCA (edgenext_modules/sdta_encoder.py:162-185); t: [B, N, C]."""
        B, N, C = t.shape
        d = C // heads
        qkv = F.linear(t, self.P(pfx + '.qkv.weight'), self.P(pfx + '.qkv.bias')).reshape(B, N, 3, heads, d)
        q, k, v = [qkv[:, :, i].permute(0, 2, 3, 1) for i in range(3)]             # [B, heads, d, N]
        q = q / q.norm(dim=-1, keepdim=True).clamp_min(1e-12)                      # F.normalize over the N tokens
        k = k / k.norm(dim=-1, keepdim=True).clamp_min(1e-12)
        attn = (q @ k.transpose(-2, -1)) * self.P(pfx + '.temperature')
        attn = attn.softmax(dim=-1)
        o = (attn @ v).permute(0, 3, 1, 2).reshape(B, N, C)
        return F.linear(o, self.P(pfx + '.proj.weight'), self.P(pfx + '.proj.bias'))

    def sdta_encoder(self, x, pfx, scales, heads):
        """SDTAEncoder (edgenext_modules/sdta_encoder.py:39-74)."""
        B, C, H, W = x.shape
        width = max(int(math.ceil(C / scales)), int(math.floor(C // scales)))
        spx = torch.split(x, width, 1)
        outs, sp = [], None
        for i in range(scales - 1):
            sp = spx[i] if i == 0 else sp + spx[i]
            sp = self.conv(sp, f'{pfx}.convs.{i}', pad=1, groups=width)
            outs.append(sp)
        y = torch.cat(outs + [spx[scales - 1]], 1)
        t = y.reshape(B, C, H * W).permute(0, 2, 1)
        if self.has(pfx + '.pos_embd.token_projection.weight'):
            t = t + self.pos_fourier(pfx + '.pos_embd', H, W).reshape(1, C, H * W).permute(0, 2, 1)
        tn = self.ln_last(t, self.P(pfx + '.norm_xca.weight'), self.P(pfx + '.norm_xca.bias'), 1e-6)
        t = t + self.P(pfx + '.gamma_xca') * self.xca(tn, pfx + '.xca', heads)
        t = self.ln_last(t, self.P(pfx + '.norm.weight'), self.P(pfx + '.norm.bias'), 1e-6)
        t = F.gelu(F.linear(t, self.P(pfx + '.pwconv1.weight'), self.P(pfx + '.pwconv1.bias')))
        t = F.linear(t, self.P(pfx + '.pwconv2.weight'), self.P(pfx + '.pwconv2.bias'))
        t = self.P(pfx + '.gamma') * t
        return x + t.reshape(B, H, W, C).permute(0, 3, 1, 2)

    @staticmethod
    def ln_first(x, w, b, eps=1e-6):
        """channels_first LayerNorm (edgenext_modules/layers.py:21-26)."""
        u = x.mean(1, keepdim=True)
        s = (x - u).pow(2).mean(1, keepdim=True)
        return w[:, None, None] * ((x - u) / torch.sqrt(s + eps)) + b[:, None, None]

    def edgenext(self, x, pfx):
        """EdgeNeXt.forward_features (edgenext_modules/edgenext.py:73-86)."""
        cfg = EDGENEXT[self.phi]
        feats = []
        for i in range(4):
            d = f'{pfx}.downsample_layers.{i}'
            if i == 0:
                x = self.conv(x, d + '.0', stride=4)
                x = self.ln_first(x, self.P(d + '.1.weight'), self.P(d + '.1.bias'))
            else:
                x = self.ln_first(x, self.P(d + '.0.weight'), self.P(d + '.0.bias'))
                x = self.conv(x, d + '.1', stride=2)
            for j in range(cfg['depths'][i]):
                b = f'{pfx}.stages.{i}.{j}'
                if i > 0 and j == cfg['depths'][i] - 1:          # global_block=[0,1,1,1]
                    x = self.sdta_encoder(x, b, cfg['scales'][i], cfg['heads'])
                else:
                    x = self.conv_encoder(x, b, cfg['ks'][i])
                x = self._b(f'backbone.s{i}.b{j}', x)
            feats.append(x)
        return feats

    # ------------------------------------------------------------------ MobileViT (a6)
    def mv_conv_bn_silu(self, x, pfx, stride=1, pad=0):
        y = self.bn(self.conv(x, pfx + '.0', stride=stride, pad=pad), pfx + '.1', 1e-5)
        return y * torch.sigmoid(y)

    def mv2block(self, x, pfx, stride, oup):
        """MV2Block, expansion != 1 (mobilevit_modules/mobilevit.py:93-131)."""
        inp = x.shape[1]
        y = self.mv_conv_bn_silu(x, pfx + '.conv', 1, 0)                           # .conv.0/.1 pw + BN + SiLU
        hid = y.shape[1]
        y = self.bn(self.conv(y, pfx + '.conv.3', stride=stride, pad=1, groups=hid), pfx + '.conv.4', 1e-5)
        y = y * torch.sigmoid(y)
        y = self.bn(self.conv(y, pfx + '.conv.6'), pfx + '.conv.7', 1e-5)
        return self._b(pfx, x + y if (stride == 1 and inp == oup) else y)

    def mv_transformer(self, t, pfx, depth, heads=4, dim_head=8):
        """Transformer / Attention / FeedForward (mobilevit.py:33-90); t: [B, P, N, D]."""
        for l in range(depth):
            a = f'{pfx}.layers.{l}.0'
            tn = self.ln_last(t, self.P(a + '.norm.weight'), self.P(a + '.norm.bias'), 1e-5)
            qkv = F.linear(tn, self.P(a + '.fn.to_qkv.weight'))
            B, Pn, N, _ = qkv.shape
            q, k, v = [z.reshape(B, Pn, N, heads, dim_head).permute(0, 1, 3, 2, 4) for z in qkv.chunk(3, dim=-1)]
            attn = (q @ k.transpose(-1, -2) * dim_head ** -0.5).softmax(dim=-1)
            o = (attn @ v).permute(0, 1, 3, 2, 4).reshape(B, Pn, N, heads * dim_head)
            t = self._b(a, F.linear(o, self.P(a + '.fn.to_out.0.weight'), self.P(a + '.fn.to_out.0.bias')) + t)
            f = f'{pfx}.layers.{l}.1'
            tn = self.ln_last(t, self.P(f + '.norm.weight'), self.P(f + '.norm.bias'), 1e-5)
            h = F.linear(tn, self.P(f + '.fn.net.0.weight'), self.P(f + '.fn.net.0.bias'))
            h = h * torch.sigmoid(h)
            t = self._b(f, F.linear(h, self.P(f + '.fn.net.3.weight'), self.P(f + '.fn.net.3.bias')) + t)
        return t

    def mvit_block(self, x, pfx, depth):
        """MobileViTBlock (mobilevit.py:147-165), 2x2 patches."""
        y = x
        x = self._b(pfx + '.conv1', self.mv_conv_bn_silu(x, pfx + '.conv1', 1, 1))
        x = self._b(pfx + '.conv2', self.mv_conv_bn_silu(x, pfx + '.conv2'))
        B, D, H, W = x.shape
        h, w = H // 2, W // 2
        # 'b d (h ph) (w pw) -> b (ph pw) (h w) d'
        t = x.reshape(B, D, h, 2, w, 2).permute(0, 3, 5, 2, 4, 1).reshape(B, 4, h * w, D)
        t = self.mv_transformer(t, pfx + '.transformer', depth)
        x = t.reshape(B, 2, 2, h, w, D).permute(0, 5, 3, 1, 4, 2).reshape(B, D, H, W)
        x = self._b(pfx + '.conv3', self.mv_conv_bn_silu(x, pfx + '.conv3'))
        x = torch.cat((x, y), 1)
        return self._b(pfx + '.conv4', self.mv_conv_bn_silu(x, pfx + '.conv4', 1, 1))

    def mobilevit(self, x, pfx):
        """MobileViT.forward (mobilevit.py:198-222)."""
        ch = MOBILEVIT[self.phi]['ch']
        x = self._b(pfx + '.conv1', self.mv_conv_bn_silu(x, pfx + '.conv1', 2, 1))
        x = self.mv2block(x, pfx + '.mv2.0', 1, ch[1])
        x = self.mv2block(x, pfx + '.mv2.1', 2, ch[2])
        x = self.mv2block(x, pfx + '.mv2.2', 1, ch[3])
        x = self.mv2block(x, pfx + '.mv2.3', 1, ch[3])
        f2 = x
        x = self.mv2block(x, pfx + '.mv2.4', 2, ch[4])
        x = self.mvit_block(x, pfx + '.mvit.0', 2)
        f3 = x
        x = self.mv2block(x, pfx + '.mv2.5', 2, ch[6])
        x = self.mvit_block(x, pfx + '.mvit.1', 4)
        f4 = x
        x = self.mv2block(x, pfx + '.mv2.6', 2, ch[8])
        x = self.mvit_block(x, pfx + '.mvit.2', 3)
        f5 = self.mv_conv_bn_silu(x, pfx + '.conv2')
        return [f2, f3, f4, f5]

    # ------------------------------------------------------------------ neck (a7-a13)
    def spp_block(self, x, pfx):
        """SPP / SPPF (neck/spp.py:41-67): Conv = conv + BN(1e-3) + SiLU."""
        def cv(z, p):
            y = self.bn(self.conv(z, p + '.conv'), p + '.bn', 1e-3)
            return y * torch.sigmoid(y)
        y = cv(x, pfx + '.cv1')
        if self.spp:
            pools = [F.max_pool2d(y, k, 1, k // 2) for k in (5, 9, 13)]
        else:
            y1 = F.max_pool2d(y, 5, 1, 2)
            y2 = F.max_pool2d(y1, 5, 1, 2)
            pools = [y1, y2, F.max_pool2d(y2, 5, 1, 2)]
        return cv(torch.cat([y] + pools, 1), pfx + '.cv2')

    def ghost_dual_fpn(self, x):
        """GhostDualFPN.forward (neck/ghostdualfpn.py:156-200)."""
        f = 'image_radar_encoder.fpn'
        w = self.w
        feats = self.edgenext(x, f + '.backbone') if self.backbone == 'en' else self.mobilevit(x, f + '.backbone')
        m2, m3, m4, m5 = feats
        m2, m3, m4, m5 = self._b('map2', m2), self._b('map3', m3), self._b('map4', m4), self._b('map5', m5)
        p5 = self.spp_block(m5, f + '.spp')
        p5 = self._b('spp', p5)
        p4 = torch.cat([self.upsample(p5, f + '.upsample_5_to_4'), m4], 1)
        p4 = self.ghost_bottleneck(p4, f + '.ghost_5_to_4', w[2])
        p3 = torch.cat([self.upsample(p4, f + '.upsample_4_to_3'), m3], 1)
        p3 = self.ghost_bottleneck(p3, f + '.ghost_4_to_3', w[1])
        p4, p3 = self._b('fpn4', p4), self._b('fpn3', p3)
        outs = {}
        for name, sa, oup in (('lane', 'stage_3_lane_seg', 2), ('se', 'stage_3_semantic_seg', self.num_seg)):
            y = self.shuffle_attention(p3, f'{f}.{sa}')
            y = self._b(f'{name}.sa', y)
            for lvl, c in (('3_to_2', w[1]), ('2_to_1', w[0]), ('1_to_0', w[0])):
                y = self.upsample(y, f'{f}.{name}_seg_{lvl}')
                y = self.ghost(y, f'{f}.{name}_seg_ghost_{lvl}', c)
                y = self._b(f'{name}.{lvl}', y)
            outs[name] = self.ghost(y, f'{f}.{name}_seg_head', oup)
        return outs['se'], outs['lane'], (p5 + m5, p4 + m4, p3 + m3)

    # ------------------------------------------------------------------ CSP neck
    def base_conv_act(self, x, pfx, act, k=1):
        """BaseConv = conv(k, pad (k-1)//2, no bias) + BN(eps 1e-3) + act (backbone/conv_utils/normal_conv.py:36-47)."""
        y = self.bn(self.conv(x, pfx + '.conv', pad=(k - 1) // 2), pfx + '.bn', 1e-3)
        return y * torch.sigmoid(y) if act == 'silu' else F.relu(y)

    def csp_bottleneck(self, x, pfx, cout):
        """Bottleneck (neck/cspdualfpn.py:42-57): 1x1 (SiLU) -> 3x3 (BaseConv default act = ReLU), + x when in == out."""
        y = self.base_conv_act(self.base_conv_act(x, pfx + '.conv1', 'silu'), pfx + '.conv2', 'relu', 3)
        return y + x if x.shape[1] == cout else y

    def csp_layer(self, x, pfx):
        """CSPLayer, n = 1 (neck/cspdualfpn.py:60-78)."""
        x1 = self.base_conv_act(x, pfx + '.conv1', 'silu')
        x2 = self.base_conv_act(x, pfx + '.conv2', 'silu')
        x1 = self.csp_bottleneck(x1, pfx + '.m.0', x1.shape[1])
        return self.base_conv_act(torch.cat((x1, x2), 1), pfx + '.conv3', 'silu')

    def csp_dual_fpn(self, x):
        """CSPDualFPN.forward (neck/cspdualfpn.py:193-237): the GDF graph with CSPLayer / Bottleneck blocks."""
        f = 'image_radar_encoder.fpn'
        feats = self.edgenext(x, f + '.backbone') if self.backbone == 'en' else self.mobilevit(x, f + '.backbone')
        m2, m3, m4, m5 = feats
        m2, m3, m4, m5 = self._b('map2', m2), self._b('map3', m3), self._b('map4', m4), self._b('map5', m5)
        p5 = self.spp_block(m5, f + '.spp')
        p5 = self._b('spp', p5)
        p4 = self.csp_layer(torch.cat([self.upsample(p5, f + '.upsample_5_to_4'), m4], 1), f + '.ghost_5_to_4')
        p3 = self.csp_layer(torch.cat([self.upsample(p4, f + '.upsample_4_to_3'), m3], 1), f + '.ghost_4_to_3')
        p4, p3 = self._b('fpn4', p4), self._b('fpn3', p3)
        outs = {}
        w = self.w
        for name, sa, oup in (('lane', 'stage_3_lane_seg', 2), ('se', 'stage_3_semantic_seg', self.num_seg)):
            y = self.shuffle_attention(p3, f'{f}.{sa}')
            y = self._b(f'{name}.sa', y)
            for lvl, c in (('3_to_2', w[1]), ('2_to_1', w[0]), ('1_to_0', w[0])):
                y = self.upsample(y, f'{f}.{name}_seg_{lvl}')
                y = self.csp_bottleneck(y, f'{f}.{name}_seg_ghost_{lvl}', c)
                y = self._b(f'{name}.{lvl}', y)
            outs[name] = self.csp_bottleneck(y, f'{f}.{name}_seg_head', oup)
        return outs['se'], outs['lane'], (p5 + m5, p4 + m4, p3 + m3)

    # ------------------------------------------------------------------ radar branch (a14-a15)
    def rc_block(self, x, pfx, down):
        """RCBlock / RadarConv / DeformableConv2d (backbone/radar/RadarEncoder.py:38-74, conv_utils/dcn.py:49-63)."""
        d = pfx + '.radar_conv.deformable_conv'
        y = F.avg_pool2d(x, 3, stride=1, padding=1)                                # count_include_pad=True
        off = self.conv(y, d + '.offset_conv', pad=1)
        msk = 2.0 * torch.sigmoid(self.conv(y, d + '.modulator_conv', pad=1))
        y = deform_conv2d(y, off, self.P(d + '.regular_conv.weight'), None, stride=(1, 1), padding=1, mask=msk)
        y = F.relu(self.bn(self.conv(y, pfx + '.weight_conv1'), pfx + '.norm', 1e-5))
        y = x + y
        return self.conv(y, pfx + '.weight_conv2', stride=2, pad=1) if down else self.conv(y, pfx + '.weight_conv2')

    def rcnet(self, x):
        """RCNet.forward (RadarEncoder.py:99-109): 8 blocks, taps after blocks 3, 5, 7."""
        down = [True, True, False, True, False, True, False, True]
        outs = []
        for i in range(8):
            x = self.rc_block(x, f'image_radar_encoder.radar_encoder.rc_blocks.{i}', down[i])
            x = self._b(f'radar.b{i}', x)
            if i > 1 and i % 2 == 1:
                outs.append(x)
        return outs

    # ------------------------------------------------------------------ fusion + head (a16-a17)
    def fuse(self, img, rad, stage):
        """IREncoder fusion (backbone/IREncoder.py:79-89)."""
        e = 'image_radar_encoder'
        z = torch.cat([self.eca(img, f'{e}.channel_attn_stage{stage}.0'),
                       self.eca(rad, f'{e}.channel_attn_stage{stage}.1')], 1)
        return F.relu(self.bn(z, f'{e}.norm_stage{stage}', 1e-5))

    def head(self, feats):
        """DecoupleHead.forward (head/decouplehead.py:58-103), nano head, depthwise."""
        outs = []
        for k, x in enumerate(feats):
            x = self.base_conv(x, f'det_head.stems.{k}')
            c = self.base_dwconv(self.base_dwconv(x, f'det_head.cls_convs.{k}.0'), f'det_head.cls_convs.{k}.1')
            r = self.base_dwconv(self.base_dwconv(x, f'det_head.reg_convs.{k}.0'), f'det_head.reg_convs.{k}.1')
            outs.append(torch.cat([self.conv(r, f'det_head.reg_preds.{k}'), self.conv(r, f'det_head.obj_preds.{k}'),
                                   self.conv(c, f'det_head.cls_preds.{k}')], 1))
        return outs

    # ------------------------------------------------------------------ PointNet (a18)
    def _c1(self, x, pfx, bn=None, relu=True):
        """conv1d k=1 (or linear) on [B, C, N] / [B, C] + optional BN1d(eps 1e-5) + ReLU."""
        w = self.P(pfx + '.weight')
        y = F.conv1d(x, w, self.P(pfx + '.bias')) if x.dim() == 3 else F.linear(x, w, self.P(pfx + '.bias'))
        if bn is not None:
            y = self.bn(y, bn, 1e-5)
        return F.relu(y) if relu else y

    def stn(self, x, pfx, k):
        """STN3d / STNkd (nets/pointcloudseg/pointnet2/pointnet_utils.py:27-45,67-85)."""
        y = self._c1(x, pfx + '.conv1', pfx + '.bn1')
        y = self._c1(y, pfx + '.conv2', pfx + '.bn2')
        y = self._c1(y, pfx + '.conv3', pfx + '.bn3')
        y = y.max(dim=2)[0]
        y = self._c1(y, pfx + '.fc1', pfx + '.bn4')
        y = self._c1(y, pfx + '.fc2', pfx + '.bn5')
        y = self._c1(y, pfx + '.fc3', None, relu=False)
        return (y + torch.eye(k, device=y.device).flatten().unsqueeze(0)).view(-1, k, k)

    def pointnet(self, pts):
        """PointNet_SEG.forward / PointNetEncoder.forward (pointnet_sem_seg.py:26-37, pointnet_utils.py:103-133)."""
        p = 'pc_seg_model'
        B, D, N = pts.shape
        trans = self.stn(pts, p + '.feat.stn', 3)
        x = pts.transpose(2, 1)
        xyz = torch.bmm(x[:, :, :3], trans)
        x = torch.cat([xyz, x[:, :, 3:]], 2).transpose(2, 1)
        x = self._c1(x, p + '.feat.conv1', p + '.feat.bn1')
        tf = self.stn(x, p + '.feat.fstn', 32)
        x = torch.bmm(x.transpose(2, 1), tf).transpose(2, 1)
        pointfeat = x
        x = self._c1(x, p + '.feat.conv2', p + '.feat.bn2')
        x = self._c1(x, p + '.feat.conv3', p + '.feat.bn3', relu=False)
        g = x.max(dim=2)[0]
        self.taps.update({'pc.trans': trans, 'pc.trans_feat': tf, 'pc.global': g})
        x = torch.cat([g.unsqueeze(2).expand(-1, -1, N), pointfeat], 1)
        x = self._c1(x, p + '.conv1', p + '.bn1')
        x = self._c1(x, p + '.conv2', p + '.bn2')
        x = self._c1(x, p + '.conv3', p + '.bn3')
        x = self._c1(x, p + '.conv4', None, relu=False)
        return F.log_softmax(x.transpose(2, 1), dim=-1)

    # ------------------------------------------------------------------ whole forward (a1)
    @torch.no_grad()
    def forward(self, x, x_radar, x_pc):
        """Achelous.forward (nets/Achelous.py:49-53)."""
        self.taps = {}
        if self.pn2 is None:
            pc = self.pointnet(x_pc.float())
        else:
            pc = torch.from_numpy(self.pn2.forward(x_pc.float()))
            self.taps.update({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in self.pn2.taps.items()})
        if self.boundary_dtype is not None:
            x, x_radar = x.to(self.boundary_dtype), x_radar.to(self.boundary_dtype)
        se, lane, (q5, q4, q3) = self.ghost_dual_fpn(x.float()) if self.neck == 'gdf' else self.csp_dual_fpn(x.float())
        r3, r4, r5 = self.rcnet(x_radar.float())
        self.taps.update({'r3': r3, 'r4': r4, 'r5': r5})
        q5, q4, q3 = self._b('q5', q5), self._b('q4', q4), self._b('q3', q3)
        p3, p4, p5 = self.fuse(q3, r3, 3), self.fuse(q4, r4, 4), self.fuse(q5, r5, 5)
        p3, p4, p5 = self._b('p3', p3), self._b('p4', p4), self._b('p5', p5)
        det = self.head((p3, p4, p5))
        return det, se, lane, pc


# ------------------------------------------------------------------------------------------- post-processing
def decode_outputs(outputs, input_shape):
    """utils/utils_bbox.py:33-85 — [B,5+C,h,w] x3 -> [B, sum(hw), 5+C] with boxes (cx,cy,w,h) normalised to [0,1]."""
    hw = [o.shape[-2:] for o in outputs]
    out = torch.cat([o.flatten(start_dim=2) for o in outputs], dim=2).permute(0, 2, 1).clone().float()
    out[:, :, 4:] = torch.sigmoid(out[:, :, 4:])
    grids, strides = [], []
    for h, w in hw:
        gy, gx = torch.meshgrid(torch.arange(h), torch.arange(w), indexing='ij')
        grids.append(torch.stack((gx, gy), 2).view(1, -1, 2).float())
        strides.append(torch.full((1, h * w, 1), input_shape[0] / h, dtype=torch.float32))
    grids, strides = torch.cat(grids, 1), torch.cat(strides, 1)
    out[..., :2] = (out[..., :2] + grids) * strides
    out[..., 2:4] = torch.exp(out[..., 2:4]) * strides
    out[..., [0, 2]] = out[..., [0, 2]] / input_shape[1]
    out[..., [1, 3]] = out[..., [1, 3]] / input_shape[0]
    return out


def non_max_suppression(prediction, num_classes, conf_thres=0.5, nms_thres=0.4):
    """utils/utils_bbox.py:87-132 up to (and excluding) the host-side un-letterboxing.
    Returns per image (rows [K,7] = x1,y1,x2,y2,obj,cls_conf,cls_id ; kept anchor indices [K]) in
    descending-score order."""
    p = prediction.detach().float().cpu().numpy().astype(np.float32).copy()
    half = np.float32(2)
    x1 = p[:, :, 0] - p[:, :, 2] / half
    y1 = p[:, :, 1] - p[:, :, 3] / half
    x2 = p[:, :, 0] + p[:, :, 2] / half
    y2 = p[:, :, 1] + p[:, :, 3] / half
    p[:, :, 0], p[:, :, 1], p[:, :, 2], p[:, :, 3] = x1, y1, x2, y2
    results = []
    for img in p:
        cls = img[:, 5:5 + num_classes]
        cid = cls.argmax(1)                       # first maximum, as torch.max
        cconf = cls[np.arange(cls.shape[0]), cid]
        score = img[:, 4] * cconf
        sel = np.where(score >= np.float32(conf_thres))[0]
        det = np.concatenate([img[sel, :5], cconf[sel, None], cid[sel, None].astype(np.float32)], 1)
        keep = batched_nms_np(det[:, :4], det[:, 4] * det[:, 5], det[:, 6], nms_thres)
        results.append((det[keep], sel[keep]))
    return results
