"""Conditioning producers on the HIP kernels: drop-ins for

  lvdm.modules.encoders.ip_resampler.Resampler                      ip_resampler.py:93-135 (image_proj_model, ddpm3d.py:683)
  lvdm.modules.encoders.condition.FrozenOpenCLIPEmbedder            condition.py:174-235  (cond_stage_model, text)
  lvdm.modules.encoders.condition.FrozenOpenCLIPImageEmbedderV2     condition.py:298-365  (embedder, image tokens)

with the reference's constructor arguments and state-dict keys.  Every computation is a kernel of libdynscaler_hip.so:
linears on ds_gemm_f16 (bias / residual epilogues), ds_layernorm, ds_attention_enc_f16 (head_dim 64 with the causal
mask for text, 80 for ViT-H/14 images, 64 for the Resampler's 273 keys), ds_gelu_f16, ds_embed_tokens, ds_clip_preprocess,
ds_patchify + GEMM for conv1, ds_vit_assemble.  Activations are rows x channels fp16 (row = token), accumulation fp32.

The pipelines call these once per distinct prompt / image crop and cache the result (sphere.py `emb_cache`,
pipelines.py per-prompt cache), so nothing here sits on the per-step path.

Not included: the BPE tokenizer.  open_clip's vocabulary file is not in this image, so `FrozenOpenCLIPEmbedder.forward`
takes token ids ([b, 77] integer tensor) or strings plus a caller-supplied `tokenizer` callable; strings without a
tokenizer raise.  open_clip itself is absent, so the towers are pinned only against an independent implementation of
the same architecture (tests/golden/make_golden.py g16); the Resampler is pinned on the reference.
"""
import torch
import torch.nn as nn

from . import ops
from .encoder_spec import (CLIP_MEAN, CLIP_STD, CLIP_VIT_H_14, clip_text_param_shapes, clip_vision_param_shapes,
                           resampler_param_shapes)


class _KeyedModule(nn.Module):
    """Parameters stored under the reference's state-dict keys; `prepare(device)` repacks them for the kernels."""

    def _init_keys(self, shapes):
        self._shapes = dict(shapes)
        self._params = nn.ParameterDict()
        for key, shape in self._shapes.items():
            self._params[key.replace(".", "/")] = nn.Parameter(torch.zeros(shape), requires_grad=False)
        self._packed, self._device = None, None

    def state_dict(self, *a, **k):
        return {key: self._params[key.replace(".", "/")].data for key in self._shapes}

    def load_state_dict(self, sd, strict=True):
        missing = [k for k in self._shapes if k not in sd]
        unexpected = [k for k in sd if k not in self._shapes and not self._ignored_key(k)]
        if strict and (missing or unexpected):
            raise RuntimeError(f"{type(self).__name__}.load_state_dict: missing {missing[:4]}, unexpected {unexpected[:4]}")
        for k in self._shapes:
            if k in sd:
                assert tuple(sd[k].shape) == tuple(self._shapes[k]), (k, tuple(sd[k].shape), self._shapes[k])
                self._params[k.replace(".", "/")].data = sd[k].detach().clone().float()
        self._packed = None
        return missing, unexpected

    def _ignored_key(self, k):
        return False

    def _prepared(self, device):
        dev = torch.device(device)
        if dev.type != "cuda":
            raise ops._lib.DsError(f"{type(self).__name__}: device {dev}; the DynamicScaler encoders have no CPU fallback")
        if self._packed is None or self._device != dev:
            sd = self.state_dict()
            self._packed = {k: (v.to(dev, torch.float16) if self._is_matrix(k, v) else v.to(dev, torch.float32)).contiguous()
                            for k, v in sd.items()}
            self._device = dev
            self._pack_extra(sd, dev)
            self.__dict__.pop("_graphs", None)      # captured graphs point at the previous weight buffers
        return self._packed

    def _is_matrix(self, k, v):
        return v.dim() == 2 and (k.endswith(".weight") or k.endswith("in_proj_weight"))

    def _pack_extra(self, sd, dev):
        pass

    def _replayed(self, fn, x):
        """fn(x) as a hipGraph replay (one graph per input signature, captured on its second use; eager before that and
        whenever capture is unavailable).  A tower is ~300 small launches on <= 273 tokens: enqueueing them from Python
        costs 4x their GPU time.  Returns a fresh tensor (callers cache embeddings)."""
        if not getattr(self, "use_graph", True) or not x.is_cuda:
            return fn(x)
        graphs = self.__dict__.setdefault("_graphs", {})
        key = (tuple(x.shape), x.dtype, x.device)
        ent = graphs.get(key)
        if ent is None:
            graphs[key] = "warm"
            return fn(x)
        if ent == "warm":
            sx = x.clone()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            multi = torch.distributed.is_available() and torch.distributed.is_initialized()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local" if multi else "global"):
                    out = fn(sx)
            except Exception as e:
                import warnings
                warnings.warn(f"hipGraph capture of {type(self).__name__} failed ({type(e).__name__}: {e}); running eagerly")
                torch.cuda.synchronize()
                self.use_graph = False
                return fn(x)
            ent = graphs[key] = (g, sx, out)
        else:
            ent[1].copy_(x)
        ent[0].replay()
        return ent[2].clone()


def _rows16(x):
    """[b, n, C] (any float dtype, on the device) -> contiguous fp16 rows [b*n, C]."""
    return x.reshape(-1, x.shape[-1]).to(torch.float16).contiguous()


def _residual_block(P, p, x, b, n, heads, causal):
    """open_clip ResidualAttentionBlock on rows x [b*n, W] (see oracle/encoders.py:_resblock)."""
    W = x.shape[1]
    hd = W // heads
    h = ops.layernorm(x, P[p + ".ln_1.weight"], P[p + ".ln_1.bias"])
    qkv = ops.gemm(h, P[p + ".attn.in_proj_weight"], P[p + ".attn.in_proj_bias"], M=b * n, N=3 * W, K=W)
    a = torch.empty((b * n, W), dtype=torch.float16, device=x.device)
    ops.attention_enc(qkv, qkv[:, W:], qkv[:, 2 * W:], a, batch=b, heads=heads, nq=n, nk=n, ldq=3 * W, ldk=3 * W,
                      ldv=3 * W, ldo=W, head_dim=hd, scale=hd ** -0.5, causal=causal)
    x = ops.gemm(a, P[p + ".attn.out_proj.weight"], P[p + ".attn.out_proj.bias"], residual=x, M=b * n, N=W, K=W)
    h = ops.layernorm(x, P[p + ".ln_2.weight"], P[p + ".ln_2.bias"])
    mlp = P[p + ".mlp.c_fc.weight"].shape[0]
    h = ops.gelu_(ops.gemm(h, P[p + ".mlp.c_fc.weight"], P[p + ".mlp.c_fc.bias"], M=b * n, N=mlp, K=W))
    return ops.gemm(h, P[p + ".mlp.c_proj.weight"], P[p + ".mlp.c_proj.bias"], residual=x, M=b * n, N=W, K=mlp)


class Resampler(_KeyedModule):
    def __init__(self, dim=1024, depth=8, dim_head=64, heads=16, num_queries=8, embedding_dim=768, output_dim=1024,
                 ff_mult=4):
        super().__init__()
        if dim_head != 64:
            raise ValueError("Resampler: the HIP attention kernels need dim_head == 64")
        self.cfg = dict(dim=dim, depth=depth, dim_head=dim_head, heads=heads, num_queries=num_queries,
                        embedding_dim=embedding_dim, output_dim=output_dim, ff_mult=ff_mult)
        self._init_keys(resampler_param_shapes(**self.cfg))

    def _pack_extra(self, sd, dev):
        self._packed["latents16"] = sd["latents"][0].to(dev, torch.float16).contiguous()

    @torch.no_grad()
    def forward(self, x):
        """x [b, n, embedding_dim] on the device -> fp16 [b, num_queries, output_dim]."""
        P = self._prepared(x.device)
        c = self.cfg
        D, nq, inner = c["dim"], c["num_queries"], c["dim_head"] * c["heads"]
        hid = int(D * c["ff_mult"])
        b, n, E = x.shape
        outs = []
        for bi in range(b):          # get_image_embeds is called with one image at a time (ddpm3d.py:689-693)
            xr = ops.gemm(_rows16(x[bi]), P["proj_in.weight"], P["proj_in.bias"], M=n, N=D, K=E)
            lat = P["latents16"]
            kv_in = torch.empty((n + nq, D), dtype=torch.float16, device=x.device)
            for i in range(c["depth"]):
                a, f = f"layers.{i}.0", f"layers.{i}.1"
                ops.layernorm(xr, P[a + ".norm1.weight"], P[a + ".norm1.bias"], out=kv_in[:n])      # cat((x, latents), -2)
                ops.layernorm(lat, P[a + ".norm2.weight"], P[a + ".norm2.bias"], out=kv_in[n:])
                q = ops.gemm(kv_in[n:], P[a + ".to_q.weight"], M=nq, N=inner, K=D)
                kv = ops.gemm(kv_in, P[a + ".to_kv.weight"], M=n + nq, N=2 * inner, K=D)
                att = torch.empty((nq, inner), dtype=torch.float16, device=x.device)
                # (q*s)(k*s)^T with s = dim_head^-1/4 (ip_resampler.py:81-83) == scale dim_head^-1/2 on the fp32 scores
                ops.attention_enc(q, kv, kv[:, inner:], att, batch=1, heads=c["heads"], nq=nq, nk=n + nq, ldq=inner,
                                  ldk=2 * inner, ldv=2 * inner, ldo=inner, head_dim=64, scale=c["dim_head"] ** -0.5)
                lat = ops.gemm(att, P[a + ".to_out.weight"], residual=lat, M=nq, N=D, K=inner)
                h = ops.layernorm(lat, P[f + ".0.weight"], P[f + ".0.bias"])
                h = ops.gelu_(ops.gemm(h, P[f + ".1.weight"], M=nq, N=hid, K=D))
                lat = ops.gemm(h, P[f + ".3.weight"], residual=lat, M=nq, N=D, K=hid)
            o = ops.gemm(lat, P["proj_out.weight"], P["proj_out.bias"], M=nq, N=c["output_dim"], K=D)
            outs.append(ops.layernorm(o, P["norm_out.weight"], P["norm_out.bias"]))
        return outs[0].unsqueeze(0) if b == 1 else torch.stack(outs, 0)


class FrozenOpenCLIPEmbedder(_KeyedModule):
    """Text tower.  `model_cfg` (the `text` section of an open_clip model config) defaults to ViT-H-14."""
    LAYERS = ["last", "penultimate"]

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", max_length=77, freeze=True,
                 layer="last", model_cfg=None, tokenizer=None, bpe_path=None):
        """tokenizer: callable(list[str]) -> LongTensor [b, 77]; or bpe_path (or $DS_CLIP_BPE_PATH): open_clip's vocabulary file
        `bpe_simple_vocab_16e6.txt.gz` for the built-in byte-level BPE (tokenizer.ClipBpeTokenizer).  The file is not shipped."""
        super().__init__()
        assert layer in self.LAYERS
        import os as _os
        bpe_path = bpe_path or _os.environ.get("DS_CLIP_BPE_PATH")
        if tokenizer is None and bpe_path:
            from .tokenizer import ClipBpeTokenizer
            tokenizer = ClipBpeTokenizer(bpe_path, context_length=max_length)
        if model_cfg is None:
            if arch != "ViT-H-14":
                raise ValueError(f"FrozenOpenCLIPEmbedder: no built-in config for arch {arch}; pass model_cfg")
            model_cfg = CLIP_VIT_H_14["text"]
        self.text = dict(model_cfg)
        if self.text["width"] // self.text["heads"] != 64:
            raise ValueError("FrozenOpenCLIPEmbedder: the HIP text attention needs head width 64")
        self.device, self.max_length, self.layer = device, max_length, layer
        self.layer_idx = 0 if layer == "last" else 1
        self.tokenizer = tokenizer
        self._init_keys(clip_text_param_shapes(self.text))

    def _ignored_key(self, k):       # what the checkpoint also carries under cond_stage_model.* and the tower never reads
        return k.startswith("model.") and k[6:].split(".")[0] in ("text_projection", "logit_scale", "attn_mask")

    def _is_matrix(self, k, v):
        return k.endswith("token_embedding.weight") or (v.dim() == 2 and k.endswith(("weight", "in_proj_weight")))

    def freeze(self):
        return self

    @torch.no_grad()
    def encode_with_transformer(self, text):
        """text: integer token ids [b, context_length] on the device -> fp16 [b, context_length, width]."""
        P = self._prepared(text.device)
        t = self.text
        b, n = text.shape
        if n != t["context_length"]:
            raise ValueError(f"encode_with_transformer: {n} tokens, context_length {t['context_length']}")
        x = ops.embed_tokens(text.to(torch.int32).contiguous(), P["model.token_embedding.weight"], P["model.positional_embedding"])
        for i in range(t["layers"] - self.layer_idx):                # condition.py:226-233: stop before the last block(s)
            x = _residual_block(P, f"model.transformer.resblocks.{i}", x, b, n, t["heads"], causal=True)
        x = ops.layernorm(x, P["model.ln_final.weight"], P["model.ln_final.bias"])
        return x.view(b, n, -1)

    def forward(self, text):
        if torch.is_tensor(text):
            tokens = text
        else:
            if self.tokenizer is None:
                raise RuntimeError("FrozenOpenCLIPEmbedder: open_clip's BPE vocabulary file is not shipped; construct with "
                                   "bpe_path=<bpe_simple_vocab_16e6.txt.gz> (or set DS_CLIP_BPE_PATH), pass "
                                   "tokenizer=callable(list[str]) -> LongTensor[b, 77], or pass token ids [b, 77]")
            tokens = self.tokenizer([text] if isinstance(text, str) else list(text))
        dev = self._params["model/positional_embedding"].device if self._device is None else self._device
        if dev.type != "cuda":
            dev = torch.device(self.device)
        return self.encode_with_transformer(tokens.to(dev))

    def encode(self, text):
        return self(text)


class FrozenOpenCLIPImageEmbedderV2(_KeyedModule):
    """Image tower: all 1 + grid^2 tokens of the last transformer block (no ln_post / proj), condition.py:336-365."""

    def __init__(self, arch="ViT-H-14", version="laion2b_s32b_b79k", device="cuda", freeze=True, layer="pooled",
                 antialias=True, model_cfg=None):
        super().__init__()
        if model_cfg is None:
            if arch != "ViT-H-14":
                raise ValueError(f"FrozenOpenCLIPImageEmbedderV2: no built-in config for arch {arch}; pass model_cfg")
            model_cfg = CLIP_VIT_H_14["vision"]
        self.vision = dict(model_cfg)
        if self.vision["head_width"] not in (64, 80):
            raise ValueError("FrozenOpenCLIPImageEmbedderV2: the HIP attention needs head width 64 or 80")
        if layer == "penultimate":
            raise NotImplementedError()                                                # as the reference (condition.py:314-316)
        self.device, self.layer, self.antialias = device, layer, antialias
        self.mean, self.std = CLIP_MEAN, CLIP_STD
        self._init_keys(clip_vision_param_shapes(self.vision))

    def _ignored_key(self, k):
        return k.startswith(("model.visual.ln_post.", "model.visual.proj")) or \
            (k.startswith("model.") and not k.startswith("model.visual."))       # the text side of the same checkpoint

    def _is_matrix(self, k, v):
        return v.dim() == 2 and k.endswith(("weight", "in_proj_weight"))

    def _pack_extra(self, sd, dev):
        w = sd["model.visual.conv1.weight"]                                    # [W, 3, P, P] -> [W][kpad], k = c*P*P + py*P + px
        k = w[0].numel()
        self._kpad = ((k + 63) // 64) * 64
        wp = torch.zeros(w.shape[0], self._kpad)
        wp[:, :k] = w.reshape(w.shape[0], -1)
        self._packed["conv1.w"] = wp.to(dev, torch.float16).contiguous()

    def freeze(self):
        return self

    @torch.no_grad()
    def preprocess(self, x):
        """x [b,3,H,W] in [-1,1] on the device -> fp32 [b,3,S,S] (condition.py:324-332)."""
        return ops.clip_preprocess(x, self.vision["image_size"], self.mean, self.std, antialias=self.antialias)

    @torch.no_grad()
    def encode_pixels(self, pix):
        """The transformer on preprocessed pixels fp32 [b,3,S,S] (condition.py:341-365)."""
        self._prepared(pix.device)
        return self._replayed(self._encode_pixels, pix.float().contiguous())

    def _encode_pixels(self, pix):
        P = self._prepared(pix.device)
        v = self.vision
        b = pix.shape[0]
        W, heads = v["width"], v["width"] // v["head_width"]
        rows = ops.patchify(pix, v["patch_size"], self._kpad)
        g2 = rows.shape[0] // b
        emb = ops.gemm(rows, P["conv1.w"], M=b * g2, N=W, K=self._kpad)
        x = ops.vit_assemble(emb, P["model.visual.class_embedding"], P["model.visual.positional_embedding"], b)
        x = ops.layernorm(x, P["model.visual.ln_pre.weight"], P["model.visual.ln_pre.bias"])
        n = g2 + 1
        for i in range(v["layers"]):
            x = _residual_block(P, f"model.visual.transformer.resblocks.{i}", x, b, n, heads, causal=False)
        return x.view(b, n, W)

    def encode_with_vision_transformer(self, x):
        return self.encode_pixels(self.preprocess(x))

    def forward(self, image, no_dropout=False):
        return self.encode_with_vision_transformer(image)

    def encode(self, image):
        return self(image)
