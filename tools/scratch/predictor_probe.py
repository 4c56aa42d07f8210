import numpy as np, torch, sys
sys.path.insert(0, '.')
import experimental_gpu_programming_for_a_spectral_numerical_integration_amd as sri

def nodes(n):
    return np.array(sri.ComputeChebyshevPoints(n)) if hasattr(sri, 'ComputeChebyshevPoints') else None

def bary_matrix(xc, xf):
    nc = len(xc)
    w = np.array([(-1.0) ** j for j in range(nc)]); w[0] *= 0.5; w[-1] *= 0.5
    P = np.zeros((len(xf), nc))
    for i, x in enumerate(xf):
        d = x - xc
        hit = np.where(d == 0)[0]
        if len(hit):
            P[i, hit[0]] = 1.0
        else:
            t = w / d
            P[i] = t / t.sum()
    return P

for scale in (1.0, 2.0, 4.0):
  for nc in (16, 32, 48):
    nf, rods = 256, 512
    rng = np.random.default_rng(1)
    qe = torch.from_numpy(rng.uniform(-scale, scale, size=(rods, 9))).cuda()
    fine = sri.RodIntegrator(nf)
    Qf = fine.integrate_quaternions(qe); itf = fine.iterations
    coarse = sri.RodIntegrator(nc)
    Qc = coarse.integrate_quaternions(qe).cpu().numpy()          # (rods, 4, mc)
    xc = np.array(sri.ComputeChebyshevPoints(nc)); xf = np.array(sri.ComputeChebyshevPoints(nf))
    P = bary_matrix(xc, xf[:-1])                                   # (mf, nc)
    q0 = np.array([1.0, 0, 0, 0])
    Qc_full = np.concatenate([Qc, np.broadcast_to(q0[None, :, None], (rods, 4, 1))], axis=2)   # (rods,4,nc)
    Q0 = np.einsum('ij,rcj->rci', P, Qc_full)                     # (rods,4,mf)
    err0 = np.abs(Q0 - Qf.cpu().numpy()).max()
    # Picard from Q0
    K = fine.strain_field(qe)
    G = fine.quaternion_rhs(K, torch.from_numpy(Q0).cuda())
    xb = torch.from_numpy(np.broadcast_to(q0, (rods, 4)).copy()).cuda()
    tips_prev = torch.from_numpy(Q0[:, :, 0].copy()).cuda()
    hist = []
    for it in range(8):
        tips = torch.empty((rods, 4), dtype=torch.float64, device='cuda')
        G = fine.quaternion_step(qe, G, xb, tips=tips)
        hist.append((tips - tips_prev).abs().max().item()); tips_prev = tips
    print(f"scale {scale} Nc {nc}: full-solve iters {itf}; interpolated guess err {err0:.2e}; tip updates {['%.1e' % h for h in hist]}")
