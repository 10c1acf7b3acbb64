"""Oracle (CPU, fp32 torch) for the Gated DeltaNet side of the hot path.

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  Every function cites the
reference lines it restates (`fla:` = /root/reference/src/llamafactory/model/fla,
`std:` = /root/reference/infinitevl/infinitevl_standard/modeling_infinitevl.py).

Layout is the reference's time-major one: q,k [B,T,H,K], v/o [B,T,H,V],
g,beta [B,T,H], state [B,H,K,V].
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F


def l2norm(x: torch.Tensor, eps: float = 1e-6, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
    """x * rsqrt(sum(x^2) + eps) over the last dim, statistics in fp32.

    fla:modules/l2norm.py:21-42 (kernel), 82-111 (host: output allocated in the
    input dtype, i.e. bf16 inputs give a bf16-rounded q_hat / k_hat).
    """
    x32 = x.float()
    y = x32 * (1.0 / torch.sqrt((x32 * x32).sum(-1, keepdim=True) + eps))
    return y if out_dtype is None else y.to(out_dtype)


def gate_math(a: torch.Tensor, b: torch.Tensor, A_log: torch.Tensor, dt_bias: torch.Tensor
              ) -> Tuple[torch.Tensor, torch.Tensor]:
    """beta = sigmoid(b) in b's dtype; g = -exp(A_log)*softplus(a + dt_bias) in fp32.

    std:1293-1294.  `a`, `b` are the a_proj / b_proj outputs [B,T,H].
    """
    beta = b.sigmoid()
    g = -A_log.float().exp() * F.softplus(a.float() + dt_bias)
    return g, beta


def gdn_recurrent(
    q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, g: torch.Tensor, beta: torch.Tensor,
    scale: Optional[float] = None, initial_state: Optional[torch.Tensor] = None,
    use_qk_l2norm_in_kernel: bool = True, qk_round_dtype: Optional[torch.dtype] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """The DEFINITION of the gated delta rule, token by token, all fp32.

    fla:ops/gated_delta_rule/fused_recurrent.py:85-101:
        S *= exp(g_t); d = beta_t*(v_t - S^T k_t); S += k_t d^T; o_t = S^T (q_t*scale)
    l2norm of q,k is applied first (fused_recurrent.py:190-192); `qk_round_dtype`
    reproduces that l2norm_fwd writes its result in the input dtype.
    Returns (o [B,T,H,V] fp32, final_state [B,H,K,V] fp32).
    """
    B, T, H, K = q.shape
    V = v.shape[-1]
    if scale is None:
        scale = K ** -0.5
    qf, kf = q.float(), k.float()
    if use_qk_l2norm_in_kernel:
        qf, kf = l2norm(qf), l2norm(kf)
        if qk_round_dtype is not None:
            qf, kf = qf.to(qk_round_dtype).float(), kf.to(qk_round_dtype).float()
    vf, gf, bf = v.float(), g.float(), beta.float()
    S = torch.zeros(B, H, K, V, dtype=torch.float32) if initial_state is None else initial_state.float().clone()
    o = torch.empty(B, T, H, V, dtype=torch.float32)
    for t in range(T):
        S = S * gf[:, t].exp()[..., None, None]
        kv = torch.einsum("bhkv,bhk->bhv", S, kf[:, t])
        d = bf[:, t][..., None] * (vf[:, t] - kv)
        S = S + torch.einsum("bhk,bhv->bhkv", kf[:, t], d)
        o[:, t] = torch.einsum("bhkv,bhk->bhv", S, qf[:, t] * scale)
    return o, S


def gdn_chunk(
    q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, g: torch.Tensor, beta: torch.Tensor,
    scale: Optional[float] = None, initial_state: Optional[torch.Tensor] = None,
    use_qk_l2norm_in_kernel: bool = True, chunk_size: int = 64,
    rounding: Optional[torch.dtype] = None, mma_rounding: Optional[torch.dtype] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Chunkwise form of the same rule (what the reference runs for T > 64).

    Restates, per (batch, head), with C = chunk_size:
      gamma  = chunk-local inclusive cumsum of g            fla:ops/utils/cumsum.py:55-56
      L      = tril(diag(beta) K K^T, -1)                   fla:ops/gated_delta_rule/wy_fast.py:164-169
      Tw     = (I+L)^-1, Tu = (I + L*Gamma)^-1              wy_fast.py:184-210
      w      = Tw diag(beta) K, u = Tu diag(beta) V         wy_fast.py:297-320
      scan:  v_new = u - (w*e^gamma) S ; S = e^{gamma_last} S + (K*e^{gamma_last-gamma})^T v_new
                                                            fla:ops/common/chunk_delta_h.py:76-120
      o      = scale*[(Q S)*e^gamma + tril((QK^T)*Gamma) v_new]   fla:ops/common/chunk_o.py:92-113
    Zero padding of a partial last chunk, gamma_last at the last valid token
    (chunk_delta_h.py:84).

    `rounding=torch.bfloat16` reproduces the reference's rounding points for a
    bf16 model (SURVEY.md section 8a'): q_hat,k_hat, beta*K, beta*V, Tw, Tu, w, u, v_new,
    the state snapshot and every dot operand rounded to bf16, fp32 accumulation,
    fp32 carried state and gamma.  `rounding=None` is exact fp32 math.

    `mma_rounding=torch.float8_e4m3fn` models the build's fp8 variant (BASELINE.json configs[4]; the reference has no
    such path): the operands of the serial pass's four products -- w e^gamma, q_hat, k_hat e^{gl-gamma}, A, the state
    snapshot and v_new -- are rounded to OCP e4m3 (clamped to +-448) instead of bf16; everything else as `rounding` says.
    Returns (o [B,T,H,V] fp32, final_state [B,H,K,V] fp32).
    """
    B, T, H, K = q.shape
    V = v.shape[-1]
    C = chunk_size
    if mma_rounding is None:
        rdm = None
    else:
        rdm = lambda x: x.clamp(-448.0, 448.0).to(mma_rounding).float()  # noqa: E731
    if scale is None:
        scale = K ** -0.5
    if rounding is None:
        rd = lambda x: x  # noqa: E731
    else:
        rd = lambda x: x.to(rounding).float()  # noqa: E731

    qf, kf = q.float(), k.float()
    if use_qk_l2norm_in_kernel:
        qf, kf = rd(l2norm(qf)), rd(l2norm(kf))
    vf, gf, bf = v.float(), g.float(), beta.float()

    NT = (T + C - 1) // C
    pad = NT * C - T
    if pad:
        qf, kf, vf = (F.pad(x, (0, 0, 0, 0, 0, pad)) for x in (qf, kf, vf))
        gf, bf = (F.pad(x, (0, 0, 0, pad)) for x in (gf, bf))

    # -> [B,H,NT,C,*]
    def chunked(x):
        return x.reshape(B, NT, C, H, -1).permute(0, 3, 1, 2, 4)

    qc, kc, vc = chunked(qf), chunked(kf), chunked(vf)
    gam = chunked(gf.unsqueeze(-1)).squeeze(-1).cumsum(-1)          # [B,H,NT,C]
    bc = chunked(bf.unsqueeze(-1))                                   # [B,H,NT,C,1]

    idx = torch.arange(C)
    lower_strict = (idx[:, None] > idx[None, :])
    lower_incl = (idx[:, None] >= idx[None, :])
    diff = gam[..., :, None] - gam[..., None, :]
    Gamma = torch.where(lower_incl, diff, torch.full_like(diff, float("-inf"))).exp()  # safe_exp, exp.py:16-18

    kb = rd(kc * bc)
    vb = rd(vc * bc)
    L = torch.where(lower_strict, kb @ kc.transpose(-1, -2), torch.zeros(()))
    eye = torch.eye(C)
    Tw = torch.linalg.solve_triangular(eye + L, eye.expand_as(L).contiguous(), upper=False)
    Tu = torch.linalg.solve_triangular(eye + L * Gamma, eye.expand_as(L).contiguous(), upper=False)
    Tw, Tu = rd(Tw), rd(Tu)
    w = rd(Tw @ kb)                                                 # [B,H,NT,C,K]
    u = rd(Tu @ vb)                                                 # [B,H,NT,C,V]

    S = torch.zeros(B, H, K, V, dtype=torch.float32) if initial_state is None else initial_state.float().clone()
    o = torch.empty(B, H, NT, C, V, dtype=torch.float32)
    for c in range(NT):
        gc = gam[:, :, c]                                           # [B,H,C]
        last = min((c + 1) * C, T) - 1 - c * C
        g_last = gc[..., last]                                      # [B,H]
        ro = rd if rdm is None else rdm                             # rounding of the products' operands
        Sr = ro(S)
        wg = ro(w[:, :, c] * gc.exp()[..., None])
        v_new = ro(u[:, :, c] - wg @ Sr)
        kd = ro(kc[:, :, c] * (g_last[..., None] - gc).exp()[..., None])
        A = ro(torch.where(lower_incl, (qc[:, :, c] @ kc[:, :, c].transpose(-1, -2)) * Gamma[:, :, c], torch.zeros(())))
        qop = qc[:, :, c] if rdm is None else rdm(qc[:, :, c])
        o[:, :, c] = ((qop @ Sr) * gc.exp()[..., None]) * scale + (A @ v_new) * scale
        S = S * g_last.exp()[..., None, None] + kd.transpose(-1, -2) @ v_new
    o = o.permute(0, 2, 3, 1, 4).reshape(B, NT * C, H, V)[:, :T]
    return o.contiguous(), S


def short_conv(
    x: torch.Tensor, weight: torch.Tensor, state: Optional[torch.Tensor] = None,
    activation: Optional[str] = "silu", bias: Optional[torch.Tensor] = None,
) -> Tuple[torch.Tensor, torch.Tensor]:
    """Causal depthwise conv1d (+SiLU) with CARRY-IN of the previous inputs.

    x [B,T,D]; weight [D,W] (nn.Conv1d weight [D,1,W] squeezed); state [B,D,W] =
    the last W raw inputs, newest LAST (fla:modules/convolution.py:236-240, 286-287).
    y[t,d] = act( sum_j weight[d,j] * ext[t+1+j,d] ), ext = concat(state^T, x).
    With state=None this is fla's prefill path (zero left context, convolution.py:
    253-266); with T==1 it is `step` (276-292).  For state given AND T>1 the
    vendored snapshot convolves with zero left context, while the pinned pip
    fla 0.4.0 carries the cached inputs in; the build follows carry-in (SURVEY.md
    Q6) -- the only semantics under which streaming 256-token frames equals one long
    prefill.  Returns (y fp32 [B,T,D], new_state fp32 [B,D,W]).
    """
    B, T, D = x.shape
    W = weight.shape[-1]
    xf = x.float()
    st = torch.zeros(B, D, W) if state is None else state.float()
    ext = torch.cat([st.transpose(1, 2), xf], dim=1)                # [B, W+T, D]
    wf = weight.float().reshape(D, W)
    y = torch.zeros(B, T, D) if bias is None else bias.float().reshape(1, 1, D).expand(B, T, D).clone()   # (convolution.py:128-160: nn.Conv1d bias)
    for j in range(W):
        y = y + ext[:, 1 + j: 1 + j + T, :] * wf[:, j]
    if activation in ("silu", "swish"):
        y = y * torch.sigmoid(y)
    new_state = ext[:, -W:, :].transpose(1, 2).contiguous()
    return y, new_state


def rmsnorm_swish_gate(x: torch.Tensor, gate: torch.Tensor, weight: Optional[torch.Tensor], eps: float = 1e-5,
                       residual: Optional[torch.Tensor] = None, return_residual: bool = False):
    """y = x*rsqrt(mean(x^2)+eps)*w * gate*sigmoid(gate), statistics in fp32.

    fla:modules/fused_norm_gate.py:27-95 with IS_RMS_NORM, ACTIVATION='swish'.  weight None: elementwise_affine=False
    (HAS_WEIGHT off).  residual: the row is x + residual in fp32 (54-58), which is also what prenorm hands back (59-60).
    """
    xf, gf = x.float(), gate.float()
    if residual is not None:
        xf = xf + residual.float()
    rstd = 1.0 / torch.sqrt((xf * xf).mean(-1, keepdim=True) + eps)
    y = xf * rstd
    if weight is not None:
        y = y * weight.float()
    y = y * gf * torch.sigmoid(gf)
    return (y, xf) if return_residual else y
