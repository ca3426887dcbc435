"""Hand-derived forward/backward of the attention decoder.  TEST INFRASTRUCTURE ONLY.

The HIP decoder kernels (asr_amd/csrc/attdec.hip) do not use autograd; they
implement the closed-form BPTT below, including the gradient that flows from
step s+1 back into step s through the alignment (alpha is not detached in
Models/decoder.py:53 and feeds the next step's conv, Models/attention.py:26).
This file spells the same algebra out in torch ops, phase by phase in the order
the kernels run them, so the derivation can be checked against autograd on the
CPU (tests/test_manual_decoder.py) before any kernel is trusted.

Shapes: hbatch (B,T,E), targets (B,L); Hd decoder width, A = 2 Hd, NC = 10
conv channels, KW = 100 taps, PAD = 50.
"""
import torch
import torch.nn.functional as F

NC, KW, PAD = 10, 100, 50


def conv_fwd(alpha, w):
    """F[b,t,c] = sum_k w[c,k] alpha[b,t+k-PAD]   (Models/attention.py:26-27)."""
    T = alpha.shape[1]
    return F.conv1d(alpha.unsqueeze(1), w, padding=PAD).transpose(1, 2)[:, :T]


def conv_bwd_alpha(dF, w):
    """d alpha_prev[b,t'] = sum_{c,k} dF[b,t'+PAD-k,c] w[c,k]."""
    B, T, _ = dF.shape
    out = dF.new_zeros(B, T)
    for k in range(KW):
        lo = max(0, k - PAD)             # t' range so that t = t'+PAD-k in [0,T)
        hi = min(T, T + k - PAD)
        if hi <= lo:
            continue
        out[:, lo:hi] += dF[:, lo + PAD - k: hi + PAD - k] @ w[:, 0, k]
    return out


def conv_bwd_w(dF, alpha_prev):
    """d w[c,k] = sum_{b,t} dF[b,t,c] alpha_prev[b,t+k-PAD]."""
    B, T, _ = dF.shape
    ap = F.pad(alpha_prev, (PAD, KW - PAD))      # ap[b, t+k] = alpha_prev[b, t+k-PAD]
    win = ap.unfold(1, KW, 1)[:, :T]             # (B,T,KW)
    return torch.einsum('btc,btk->ck', dF, win).unsqueeze(1)


def forward(P, hbatch, lengths, targets):
    """P: dict of decoder parameters (names as in the state_dict, without 'decoder.')."""
    B, T, E = hbatch.shape
    L = targets.shape[1]
    Hd = P['L_sy.weight'].shape[0]
    mask = (torch.arange(T).unsqueeze(0) < lengths.unsqueeze(1)).to(hbatch.dtype)
    PH = hbatch @ P['att.L_he.weight'].t() + P['att.L_he.bias']          # hoisted
    wee = P['att.L_ee.weight'][0]
    S = [hbatch.new_zeros(B, Hd)]
    C = [hbatch.new_zeros(B, Hd)]
    alpha_prev = hbatch.new_zeros(B, T)
    sv = dict(PH=PH, mask=mask, S=S, C=C, alpha=[], Fc=[], sp=[], g=[], u=[], gates=[])
    for s in range(L):
        Fc = conv_fwd(alpha_prev, P['att.F_conv1d.weight'])
        sp = S[s] @ P['att.L_se.weight'].t()
        th = torch.tanh(sp.unsqueeze(1) + PH + Fc @ P['att.L_fe.weight'].t())
        e = th @ wee
        p = mask * torch.exp(e - e.max(1, keepdim=True)[0])
        alpha = p / p.sum(1, keepdim=True)
        g = (alpha.unsqueeze(2) * hbatch).sum(1)
        u = torch.tanh(g @ P['L_gy.weight'].t() + P['L_gy.bias'] + S[s] @ P['L_sy.weight'].t())
        rec = (P['L_ys.weight'][targets[:, s]] + S[s] @ P['L_ss.weight'].t()
               + g @ P['L_gs.weight'].t() + P['L_gs.bias'])
        i, f, gg, o = rec.chunk(4, 1)
        i = 0.5 * torch.tanh(0.5 * i) + 0.5
        f = 0.5 * torch.tanh(0.5 * f) + 0.5
        gg = torch.tanh(gg)
        o = 0.5 * torch.tanh(0.5 * o) + 0.5
        c_new = f * C[s] + i * gg
        S.append(o * torch.tanh(c_new))
        C.append(c_new)
        for k, v in (('alpha', alpha), ('Fc', Fc), ('sp', sp), ('g', g), ('u', u),
                     ('gates', torch.cat([i, f, gg, o], 1))):
            sv[k].append(v)
        alpha_prev = alpha
    U = torch.stack(sv['u'], 1)                                           # (B,L,Hd)
    Y = U @ P['L_yy.weight'].t() + P['L_yy.bias']                          # hoisted
    return Y, sv


def backward(P, hbatch, targets, sv, dY):
    B, T, E = hbatch.shape
    L = targets.shape[1]
    Hd = P['L_sy.weight'].shape[0]
    wee = P['att.L_ee.weight'][0]
    PH, S, C = sv['PH'], sv['S'], sv['C']
    G = {k: torch.zeros_like(v) for k, v in P.items()}
    dU = dY @ P['L_yy.weight']
    G['L_yy.weight'] = dY.reshape(B * L, -1).t() @ torch.stack(sv['u'], 1).reshape(B * L, Hd)
    G['L_yy.bias'] = dY.sum((0, 1))
    dS = hbatch.new_zeros(B, Hd)
    dC = hbatch.new_zeros(B, Hd)
    dF_next = None
    dPH = torch.zeros_like(PH)
    dh = torch.zeros_like(hbatch)
    for s in range(L - 1, -1, -1):
        i, f, gg, o = sv['gates'][s].chunk(4, 1)
        tc = torch.tanh(C[s + 1])
        do = dS * tc
        dc = dS * o * (1 - tc * tc) + dC
        drec = torch.cat([dc * gg * i * (1 - i), dc * C[s] * f * (1 - f),
                          dc * i * (1 - gg * gg), do * o * (1 - o)], 1)
        dC = dc * f
        u = sv['u'][s]
        dyp = dU[:, s] * (1 - u * u)
        # small-GEMM backward
        dg = drec @ P['L_gs.weight'] + dyp @ P['L_gy.weight']
        dS_prev = drec @ P['L_ss.weight'] + dyp @ P['L_sy.weight']
        # attention backward, phase a: d alpha and the softmax dot
        alpha = sv['alpha'][s]
        dalpha = torch.einsum('be,bte->bt', dg, hbatch)
        if dF_next is not None:
            dalpha = dalpha + conv_bwd_alpha(dF_next, P['att.F_conv1d.weight'])
        dot = (alpha * dalpha).sum(1, keepdim=True)
        # phase b: energies
        de = alpha * (dalpha - dot)
        Fc = sv['Fc'][s]
        th = torch.tanh(sv['sp'][s].unsqueeze(1) + PH + Fc @ P['att.L_fe.weight'].t())
        dpre = de.unsqueeze(2) * wee * (1 - th * th)
        dsp = dpre.sum(1)
        dF = dpre @ P['att.L_fe.weight']
        dS_prev = dS_prev + dsp @ P['att.L_se.weight']
        # accumulations (hoisted out of the loop in the kernels where linear)
        dPH += dpre
        G['att.L_ee.weight'] += torch.einsum('bt,bta->a', de, th).unsqueeze(0)
        G['att.L_fe.weight'] += torch.einsum('bta,btc->ac', dpre, Fc)
        if s > 0:
            G['att.F_conv1d.weight'] += conv_bwd_w(dF, sv['alpha'][s - 1])
        G['att.L_se.weight'] += dsp.t() @ S[s]
        G['L_gs.weight'] += drec.t() @ sv['g'][s]
        G['L_ss.weight'] += drec.t() @ S[s]
        G['L_gs.bias'] += drec.sum(0)
        G['L_gy.weight'] += dyp.t() @ sv['g'][s]
        G['L_sy.weight'] += dyp.t() @ S[s]
        G['L_gy.bias'] += dyp.sum(0)
        G['L_ys.weight'].index_add_(0, targets[:, s], drec)
        dh += alpha.unsqueeze(2) * dg.unsqueeze(1)
        dS, dF_next = dS_prev, dF
    dh += dPH @ P['att.L_he.weight']
    G['att.L_he.weight'] = dPH.reshape(B * T, -1).t() @ hbatch.reshape(B * T, E)
    G['att.L_he.bias'] = dPH.sum((0, 1))
    return dh, G
