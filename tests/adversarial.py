"""TEST INFRASTRUCTURE.  Synthetic states that force the branches the smooth freestream-plus-noise states of adflow_amd.synth
never take (round-1 verdict, "parity gaps" 1), each with a host-side recount of the branch from the very arrays the reference
is run on, so a test can assert "this branch fired" before comparing:

  shock_block     Mach-3 normal shock across i (+ a uniform upstream region and an expansion ramp):
                    scalar / matrix JST  dis2 = fis2 * min(0.25, sensor) clip, dis4 = max(fis4 - dis2, 0) clip  (fluxes.F90:1228-1232, 549-551)
                    Roe                  entropy fix |lambda| < 2 eta (fluxes.F90:2432-2436)
                    MUSCL                limiter cut-off r < 0 and the epsLim clamp of a zero difference (fluxes.F90:2167-2190)
  vacuum_pocket   cells whose density / pressure are driven below 1e-4 of the free stream by one Runge-Kutta stage
                  (smoothers.F90:326, 342) and a wall whose linear pressure extrapolation turns negative (BCRoutines.F90:560-562)
  sa_extremes     Spalart-Allmaras: nuTilde -> 1e-12 nu, d2Wall -> 1e-8, pockets of uniform velocity with chi ~ 5:
                    sst = max(sst, 1e-10), rr = min(rr, 10), ft2 ~ ct3, fv2 < 0 (sa.F90:262-276)
"""
import numpy as np

from adflow_amd.params import FlowParams
from adflow_amd.synth import make_block


def _energy(blk, prm):
    w = blk["w"]
    w[..., 4] = blk["p"] / (prm.gammaConstant - 1.0) + 0.5 * w[..., 0] * (w[..., 1] ** 2 + w[..., 2] ** 2 + w[..., 3] ** 2)


def shock_block(dims, prm: FlowParams, seed=1, mach=3.0, **mk):
    """State: uniform supersonic flow along +i for i < ~40 %, a normal shock (Rankine-Hugoniot) there, a smooth expansion
    behind it, 1 % noise on the post-shock part only (so the upstream differences are EXACTLY zero)."""
    blk = make_block(*dims, prm, seed=seed, **mk)
    g = prm.gammaConstant
    w, p = blk["w"], blk["p"]
    ni = w.shape[0]
    rho1, p1 = 1.0, 1.0 / g                      # a1 = 1
    u1 = mach
    r = (g + 1) * mach ** 2 / ((g - 1) * mach ** 2 + 2)
    rho2, u2, p2 = rho1 * r, u1 / r, p1 * (1 + 2 * g / (g + 1) * (mach ** 2 - 1))
    ish = max(3, int(0.4 * ni))
    rng = np.random.default_rng(seed)
    i = np.arange(ni)[:, None, None]
    up = i < ish
    ramp = 1.0 + 0.3 * np.clip((i - ish) / max(1, ni - ish), 0, 1)          # expansion behind the shock
    noise = 1.0 + 0.01 * rng.uniform(-1, 1, p.shape)
    w[..., 0] = np.where(up, rho1, rho2 / ramp * noise)
    w[..., 1] = np.where(up, u1, u2 * ramp)
    w[..., 2] = np.where(up, 0.0, 0.05 * u2 * (noise - 1.0) * 20)
    w[..., 3] = np.where(up, 0.0, -0.03 * u2 * (noise - 1.0) * 20)
    p[...] = np.where(up, p1, p2 / ramp ** g * noise)
    _energy(blk, prm)
    return blk


def clamp_block(dims, prm: FlowParams, seed=1, **mk):
    """State for the epsLim clamp of the MUSCL limiters (fluxes.F90:2103-2294): the lower half in i is a uniform flow with
    perturbations of 2e-11 .. 2e-10 relative (differences of either sign, on both sides of epsLim = 1e-10, some exactly zero), the
    upper half carries 1 % noise, so that the residual the comparison is scaled by is of order one."""
    blk = make_block(*dims, prm, seed=seed, **mk)
    w, p = blk["w"], blk["p"]
    ni = w.shape[0]
    rng = np.random.default_rng(seed)
    i = np.arange(ni)[:, None, None]
    low = i < ni // 2
    tiny = rng.choice([0.0, 2.e-11, 5.e-11, 1.e-10, 2.e-10], size=p.shape) * rng.uniform(-1, 1, p.shape)
    base = [1.0, 0.8, 0.05, -0.03]
    for l in range(4):
        noisy = base[l] * (1.0 + 0.01 * rng.uniform(-1, 1, p.shape))
        tl = rng.choice([0.0, 2.e-11, 5.e-11, 1.e-10, 2.e-10], size=p.shape) * rng.uniform(-1, 1, p.shape)
        w[..., l] = np.where(low, base[l] + tl, noisy)
    p0 = 1.0 / prm.gammaConstant
    p[...] = np.where(low, p0 + tiny, p0 * (1.0 + 0.01 * rng.uniform(-1, 1, p.shape)))
    _energy(blk, prm)
    return blk


def count_clamped_differences(blk):
    """cells whose smaller one-sided difference (any of rho, u, v, w, p; any direction) lies strictly between 0 and epsLim"""
    q = [blk["w"][..., l] for l in range(4)] + [blk["p"]]
    n = 0
    for a in q:
        for ax in range(3):
            d = np.abs(np.diff(a, axis=ax))
            lo = [slice(None)] * 3
            hi = [slice(None)] * 3
            lo[ax] = slice(0, -1)
            hi[ax] = slice(1, None)
            m = np.minimum(d[tuple(lo)], d[tuple(hi)])
            n += int(((m > 0.0) & (m < 1.e-10)).sum())
    return n


def count_shock_branches(blk, prm: FlowParams):
    """recount of the branches on the i faces (cells 1..ie, faces i = 1..il) from the block's arrays"""
    g = prm.gammaConstant
    w, p = blk["w"], blk["p"]
    il, jl, kl = blk.il, blk.jl, blk.kl
    s = (slice(None), slice(2, jl + 1), slice(2, kl + 1))
    pp = p[s]
    plim = 0.001 * prm.pInfCorr
    dss = np.abs((pp[2:] - 2 * pp[1:-1] + pp[:-2]) / (pp[2:] + 2 * pp[1:-1] + pp[:-2] + plim))      # cells 1..ib-1
    sens = np.maximum(dss[:-1], dss[1:])                                                          # faces (i | i+1), i = 1..
    dis2_clip = int((sens[:il] > 0.25).sum())
    dis4_zero = int((prm.vis4 - prm.vis2 * np.minimum(0.25, sens[:il]) < 0).sum())
    # Roe entropy fix with the cell states as left / right (first-order recount)
    rho, u = w[s + (0,)], w[s + (1,)]
    c = np.sqrt(g * pp / rho)
    eta = 0.5 * (np.abs(u[:-1] - u[1:]) + np.abs(c[:-1] - c[1:]))
    zl, zr = np.sqrt(rho[:-1]), np.sqrt(rho[1:])
    ua = (zl * u[:-1] + zr * u[1:]) / (zl + zr)
    aa = 0.5 * (c[:-1] + c[1:])
    fix = int(((np.abs(ua - aa) < 2 * eta) | (np.abs(ua) < 2 * eta) | (np.abs(ua + aa) < 2 * eta))[1:il + 1].sum())
    # limiter: exactly-zero differences (epsLim clamp) and sign changes (cut-off at 0) of the density
    d = rho[1:] - rho[:-1]
    clamp = int((np.abs(d[:il + 1]) < 1e-10).sum())
    cutoff = int((d[:-1] * d[1:] < 0)[:il].sum())
    return dict(dis2_clip=dis2_clip, dis4_zero=dis4_zero, entropy_fix=fix, limiter_clamp=clamp, limiter_cutoff=cutoff)


def vacuum_pocket(blk, prm: FlowParams, frac=2e-4):
    """lower density and pressure of a pocket in the middle of the block to `frac` of their values (in place)"""
    nx, ny, nz = blk.nx, blk.ny, blk.nz
    s = (slice(2 + nx // 3, 2 + max(nx // 3 + 1, 2 * nx // 3)), slice(2 + ny // 3, 2 + max(ny // 3 + 1, 2 * ny // 3)),
         slice(2 + nz // 3, 2 + max(nz // 3 + 1, 2 * nz // 3)))
    blk["w"][s + (0,)] *= frac
    blk["p"][s] *= frac
    _energy(blk, prm)
    return s


def sa_extremes(dims, prm: FlowParams, seed=1, **mk):
    """RANS block: near-wall layer with nuTilde = 1e-12 nu and d2Wall = 1e-8, a pocket of UNIFORM velocity with chi ~ 5 and a
    small wall distance (negative fv2 term larger than the strain: sst clipped at 1e-10), a pocket with a huge nuTilde
    (rr clipped at 10)."""
    blk = make_block(*dims, prm, seed=seed, **mk)
    w = blk["w"]
    nu = blk["rlv"] / w[..., 0]
    nk = w.shape[2]
    k1 = 2 + max(1, blk.nz // 4)
    w[:, :, :k1, 5] = 1e-12 * nu[:, :, :k1]
    blk["d2Wall"][:, :, :max(1, blk.nz // 4)] = 1e-8
    i0, i1 = 2 + blk.nx // 3, 2 + max(blk.nx // 3 + 2, 2 * blk.nx // 3)
    k2 = min(nk - 2, k1 + max(2, blk.nz // 3))
    pocket = (slice(i0 - 2, i1 + 2), slice(None), slice(k1, k2))
    for l in (1, 2, 3):
        w[pocket + (l,)] = w[i0, 2, k1, l]                      # uniform velocity: zero strain / vorticity inside
    w[pocket + (5,)] = 5.0 * nu[pocket]
    blk["d2Wall"][i0 - 2:i1 - 2, :, k1 - 2:k2 - 2] = 1e-3
    w[:, :, k2:, 5] = 1e4 * nu[:, :, k2:]                          # rr = nuTilde / (sst kappa^2 d^2) >> 10 with d ~ 0.1 .. 1 ? set d small as well
    blk["d2Wall"][:, :, k2 - 2:] = np.minimum(blk["d2Wall"][:, :, k2 - 2:], 5e-3)
    _energy(blk, prm)
    return blk


def count_sa_branches(blk, prm: FlowParams):
    """recount of the limiters of saSource (sa.F90:230-290) on the owned cells, with the strain magnitude taken as the
    vorticity-free lower bound ss >= 0 replaced by its actual first-order estimate from the i-differences only (a recount
    for "the branch fired", not a second implementation)"""
    w = blk["w"]
    s = (slice(2, blk.il + 1), slice(2, blk.jl + 1), slice(2, blk.kl + 1))
    rho, nut = w[s + (0,)], w[s + (5,)]
    nu = blk["rlv"][s] / rho
    d2 = blk["d2Wall"]
    chi = nut / nu
    cv13 = prm.SAcv1 ** 3
    fv1 = chi ** 3 / (chi ** 3 + cv13)
    fv2 = 1.0 - chi / (1.0 + chi * fv1)
    kar2 = prm.SAKappa ** 2
    term = nut * fv2 / (kar2 * d2 * d2)
    # velocity differences over the six neighbours as a bound of the strain
    g = 0.0
    for l in (1, 2, 3):
        q = w[..., l]
        g = g + np.abs(q[3:blk.il + 2, 2:blk.jl + 1, 2:blk.kl + 1] - q[1:blk.il, 2:blk.jl + 1, 2:blk.kl + 1]) \
            + np.abs(q[2:blk.il + 1, 3:blk.jl + 2, 2:blk.kl + 1] - q[2:blk.il + 1, 1:blk.jl, 2:blk.kl + 1]) \
            + np.abs(q[2:blk.il + 1, 2:blk.jl + 1, 3:blk.kl + 2] - q[2:blk.il + 1, 2:blk.jl + 1, 1:blk.kl])
    uniform = g == 0.0
    sst_clip = int((uniform & (term < 0)).sum())                     # ss = 0 there: sst = term < 0 -> clipped at 1e-10
    rr_clip = int((uniform & (nut / (kar2 * d2 * d2) / 1e-10 > 10)).sum()) + int(((term > 0) & (fv2 > 0) & (1.0 / fv2 > 10) & uniform).sum())
    tiny_chi = int((chi < 1e-10).sum())
    fv2_neg = int((fv2 < 0).sum())
    return dict(sst_clip=sst_clip, rr_clip=rr_clip, tiny_chi=tiny_chi, fv2_negative=fv2_neg)
