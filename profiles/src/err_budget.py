"""CPU emulation of the engine's bf16 rounding points inside the fp32 oracle, to rank their contribution to log-prob error."""
import sys, numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, '.')
from oracle import model as om, vision_index as vi
R = lambda x: x.to(torch.bfloat16).float()
def lm(w, cfg, emb, pos, flags, nlayers=None):
    t = cfg.text; B, S, H = emb.shape; nh, nkv = t.num_attention_heads, t.num_key_value_heads; hd = H // nh
    cos, sin = om.mrope_cos_sin(cfg, pos); cos, sin = cos.unsqueeze(1), sin.unsqueeze(1)
    bias = torch.zeros(S, S).masked_fill(~torch.tril(torch.ones(S, S, dtype=torch.bool)), float('-inf'))
    x = emb.float()
    for li in range(t.num_hidden_layers):
        p = f"model.layers.{li}."
        h = om.rms_norm(x, w[p + "input_layernorm.weight"], t.rms_norm_eps)
        if 'xn' in flags: h = R(h)
        q = (h @ w[p + "self_attn.q_proj.weight"].t() + w[p + "self_attn.q_proj.bias"]).view(B, S, nh, hd).transpose(1, 2)
        k = (h @ w[p + "self_attn.k_proj.weight"].t() + w[p + "self_attn.k_proj.bias"]).view(B, S, nkv, hd).transpose(1, 2)
        v = (h @ w[p + "self_attn.v_proj.weight"].t() + w[p + "self_attn.v_proj.bias"]).view(B, S, nkv, hd).transpose(1, 2)
        q = q * cos + om.rotate_half(q) * sin; k = k * cos + om.rotate_half(k) * sin
        if 'qkv' in flags: q, k, v = R(q), R(k), R(v)
        kr = k.repeat_interleave(nh // nkv, 1); vr = v.repeat_interleave(nh // nkv, 1)
        att = ((q @ kr.transpose(2, 3)) * hd ** -0.5 + bias).softmax(-1)
        if 'p' in flags:
            # engine: unnormalised p=exp(s-m) rounded to bf16, normalised after PV in fp32
            s = (q @ kr.transpose(2, 3)) * hd ** -0.5 + bias
            pm = torch.exp(s - s.max(-1, keepdim=True)[0]); o = (R(pm) @ vr) / pm.sum(-1, keepdim=True)
        else:
            o = att @ vr
        o = o.transpose(1, 2).reshape(B, S, nh * hd)
        if 'ao' in flags: o = R(o)
        x = x + o @ w[p + "self_attn.o_proj.weight"].t()
        h = om.rms_norm(x, w[p + "post_attention_layernorm.weight"], t.rms_norm_eps)
        if 'xn' in flags: h = R(h)
        a = F.silu(h @ w[p + "mlp.gate_proj.weight"].t()) * (h @ w[p + "mlp.up_proj.weight"].t())
        if 'act' in flags: a = R(a)
        x = x + a @ w[p + "mlp.down_proj.weight"].t()
    h = om.rms_norm(x, w["model.norm.weight"], t.rms_norm_eps)
    if 'fn' in flags: h = R(h)
    return h @ w["lm_head.weight"].t()
def run(cfg, seed, S=96, C=32):
    w = om.make_weights(cfg, seed=seed)
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(10, 2000, (1, S), generator=g)
    emb = w["model.embed_tokens.weight"][ids]
    pos = torch.arange(S)[None, None].expand(3, 1, S)
    def logp(flags):
        lg = lm(w, cfg, emb, pos, flags)[:, :-1][:, -C:] / 0.9
        return om.selective_log_softmax(lg, ids[:, 1:][:, -C:])[0]
    ref = logp(set())
    out = {}
    for f in (['xn'], ['qkv'], ['p'], ['ao'], ['act'], ['fn'], ['xn', 'qkv', 'p', 'ao', 'act', 'fn']):
        e = (logp(set(f)) - ref).abs() / ref.abs().clamp(min=1e-2)
        out['+'.join(f)] = (e.max().item(), e.mean().item())
    return out
for L in (2, 8, 28):
    cfg = om.cfg_tiny(2); cfg.text.num_hidden_layers = L
    res = {}
    for seed in range(3):
        for k, v in run(cfg, seed).items():
            res.setdefault(k, []).append(v)
    print(f"layers={L}")
    for k, v in res.items():
        print(f"   {k:24s} max {max(a for a,_ in v):.2e}  mean {np.mean([b for _,b in v]):.2e}")
