"""Batch-shared TPS constants (fiducial lattice C, pixel centres P, inverse of Delta_C, RBF table).

Built once per module in float64 NumPy and cast to fp32, exactly as the reference does in its
constructors, so that freshly constructed modules carry the same buffers as the reference's
(`GridGenerator.inv_delta_C / P_hat`, tps_preprocessor.py:179-188,197-268;
`Attention_Enhanced_TPS.hat_C / P_hat`, tps_pp.py:357-366,368-465).  Checkpoints overwrite
them through `load_state_dict` anyway.  Host-side, one-off: not part of the timed path.
"""
import numpy as np

EPS = 1e-6   # tps_preprocessor.py:175, tps_pp.py:340


def _pair_norm(A, B):
    """|A[i] - B[j]| for (na,2), (nb,2) float64 inputs -> (na, nb)."""
    d = A[:, None, :] - B[None, :, :]
    return np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1])


def inverse_delta_C(C):
    """inv([[1, C, R], [0, C^T], [0, 1^T]]),  R = rho^2 ln(rho), rho = 1 on the diagonal."""
    F = C.shape[0]
    rho = _pair_norm(C, C)
    rho[np.arange(F), np.arange(F)] = 1.0
    R = (rho ** 2) * np.log(rho)
    delta = np.zeros((F + 3, F + 3), dtype=np.float64)
    delta[:F, 0] = 1.0
    delta[:F, 1:3] = C
    delta[:F, 3:] = R
    delta[F:F + 2, 3:] = C.T
    delta[F + 2, 3:] = 1.0
    return np.linalg.inv(delta)


def rbf_table(C, P):
    """(n, F): d^2 ln(d + eps) with d = |P[n] - C[k]|."""
    d = _pair_norm(P, C)
    return np.square(d) * np.log(d + EPS)


# ---- classic RARE TPS-STN --------------------------------------------------------------------
def classic_C(num_fiducial):
    half = int(num_fiducial / 2)
    x = np.linspace(-1.0, 1.0, half)
    return np.concatenate([np.stack([x, -np.ones(half)], axis=1),
                           np.stack([x, np.ones(half)], axis=1)], axis=0)


def classic_P(Hr, Wr):
    gx = (np.arange(-Wr, Wr, 2) + 1.0) / Wr
    gy = (np.arange(-Hr, Hr, 2) + 1.0) / Hr
    return np.stack(np.meshgrid(gx, gy), axis=2).reshape([-1, 2])


def classic(num_fiducial, rectified_img_size):
    Hr, Wr = rectified_img_size
    C, P = classic_C(num_fiducial), classic_P(Hr, Wr)
    P_hat = np.concatenate([np.ones((P.shape[0], 1)), P, rbf_table(C, P)], axis=1)
    return dict(C=C, P=P, inv_delta_C=inverse_delta_C(C).astype(np.float32),
                P_hat=P_hat.astype(np.float32))


def classic_initial_ctrl(num_fiducial):
    """Initial bias of LocalizationNetwork.localization_fc2 (tps_preprocessor.py:130-140)."""
    half = int(num_fiducial / 2)
    x = np.linspace(-1.0, 1.0, half)
    top = np.stack([x, np.linspace(0.0, -1.0, num=half)], axis=1)
    bot = np.stack([x, np.linspace(1.0, 0.0, num=half)], axis=1)
    return np.concatenate([top, bot], axis=0).astype(np.float32)


def classic_identity_ctrl(num_fiducial):
    """C' = C: the control points for which the TPS is the identity (bench workload base)."""
    return classic_C(num_fiducial).astype(np.float32)


# ---- TPS_PP / Attention_Enhanced_TPS -----------------------------------------------------------
def tpspp_C(point_size):
    py, px = point_size
    cx = np.linspace(0.5, px - 0.5, num=int(px)) / px
    cy = np.linspace(0.5, py - 0.5, num=int(py)) / py
    return np.stack(np.meshgrid(cx, cy), axis=2).reshape([-1, 2])


def tpspp_P(Hr, Wr):
    gx = np.linspace(0.5, Wr - 0.5, num=int(Wr)) / Wr
    gy = np.linspace(0.5, Hr - 0.5, num=int(Hr)) / Hr
    return np.stack(np.meshgrid(gx, gy), axis=2).reshape([-1, 2])


def tpspp(rectified_img_size, point_size):
    Hr, Wr = rectified_img_size
    C, P = tpspp_C(point_size), tpspp_P(Hr, Wr)
    return dict(C=C, P=P, hat_C=inverse_delta_C(C).astype(np.float32),
                P_hat=rbf_table(C, P).astype(np.float32), P_xy=P.astype(np.float32))


def tpspp_initial_ctrl(point_size):
    """Initial bias of TPE.localization_fc2 (tps_pp.py:279-285)."""
    py, px = point_size
    x = np.linspace(0.1, px - 0.1, num=int(px)) / px
    y = np.linspace(0.1, py - 0.1, num=int(py)) / py
    return np.stack(np.meshgrid(x, y), axis=2).reshape(-1, 2).astype(np.float32)
