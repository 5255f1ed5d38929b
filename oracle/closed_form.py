"""ORACLE (test infrastructure, never shipped, never on the product path).

Closed-form fp32 NumPy restatement of ONE mask-optimisation iteration: forward,
analytic backward and Adam.  This is the stage-by-stage specification the HIP
kernels implement; every intermediate the kernels write to
HBM is exposed so each kernel can be unit-tested against its own stage.

It is pinned against the reference through `oracle/reference_restatement.py`
(torch autograd, bit-identical to /root/reference on the golden fixtures):
tests/test_oracle_closed_form.py requires the two to agree to fp32 round-off.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this module.

Reference lines restated (relative to /root/reference):
  explainer/explain.py:665-678 (masked adjacency), :685-715 (forward), :740-808 (loss),
  models.py:58-80, 230-267, 269-316, 363-376 (encoder), torch.optim.Adam via
  utils/train_utils.py:9-10 (lr, betas (0.9, 0.999), eps 1e-8, no weight decay).
"""
import numpy as np

F32 = np.float32
C_SIZE, C_FEAT_SIZE, C_ENT, C_LAP = F32(0.005), F32(1.0), F32(1.0), F32(1.0)
BETA1, BETA2, EPS = 0.9, 0.999, 1e-8
NORM_EPS = F32(1e-12)   # F.normalize eps
BN_EPS = F32(1e-5)      # nn.BatchNorm1d default eps (apply_bn, models.py:222-228)


def sigmoid(x):
    x = x.astype(F32)
    return (F32(1) / (F32(1) + np.exp(-x, dtype=F32))).astype(F32)


class Weights:
    """fp32 copies of the encoder parameters (state_dict names of models.py)."""

    def __init__(self, sd):
        g = lambda k: np.ascontiguousarray(np.asarray(sd[k], dtype=F32))
        self.W = [g("conv_first.weight"), g("conv_block.0.weight"), g("conv_last.weight")]
        self.b = [g("conv_first.bias"), g("conv_block.0.bias"), g("conv_last.bias")]
        self.Wp = g("pred_model.weight")   # [C, H+H+O]
        self.bp = g("pred_model.bias")


def masked_adj(M, A):
    """Abar = A * (sigma(M) + sigma(M)^T)/2 with zero diagonal; also returns S = sigma(M)."""
    S = sigmoid(M)
    T = (S + S.T) / F32(2)
    Abar = (A * T).astype(F32)
    np.fill_diagonal(Abar, 0)
    return Abar, S


def batch_norm_rows(a):
    """apply_bn (models.py:222-228) on a [1, n, d] activation: a FRESH BatchNorm1d(n) in training mode, i.e. every node
    (= channel) is standardised over its d features with the biased variance, weight 1, bias 0.  -> (x_hat, 1 / std)"""
    mu = a.mean(1, dtype=F32)[:, None]
    var = ((a - mu) ** 2).mean(1, dtype=F32)[:, None]
    rs = (F32(1) / np.sqrt(var + BN_EPS)).astype(F32)
    return ((a - mu) * rs).astype(F32), rs


def forward(Abar, X0, wts, bn=False):
    """3 GraphConv layers. Returns per-layer (U_l = normalised pre-activation, r_l = row norm)."""
    U, R, Xin, RS = [], [], [X0], []
    x = X0
    for l in range(3):
        z = (Abar @ x).astype(F32)
        y = (z @ wts.W[l] + wts.b[l]).astype(F32)
        r = np.maximum(np.sqrt((y * y).sum(1, dtype=F32)), NORM_EPS).astype(F32)
        u = (y / r[:, None]).astype(F32)
        U.append(u)
        R.append(r)
        x = np.maximum(u, 0) if l < 2 else u
        if bn and l < 2:                      # gcn_forward (models.py:241-253): ReLU, then apply_bn
            x, rs = batch_norm_rows(x)
            RS.append(rs)
        if l < 2:
            Xin.append(x)
    if bn:
        return U, R, Xin, RS
    return U, R, Xin   # Xin = [X0, X1, X2] (inputs of the three layers)


def head_node(U, wts, t, y_gt, Xin=None):
    """Logits/softmax of row t, -log p[y_gt], and the direct gradient dE[t] (length H+H+O).
    Xin: the layer inputs [X0, X1, X2] when they are not relu(U) (batch norm)."""
    a1, a2 = (np.maximum(U[0][t], 0), np.maximum(U[1][t], 0)) if Xin is None else (Xin[1][t], Xin[2][t])
    e = np.concatenate([a1, a2, U[2][t]]).astype(F32)
    z = (wts.Wp @ e + wts.bp).astype(F32)
    z = z - z.max()
    p = np.exp(z, dtype=F32)
    p = (p / p.sum(dtype=F32)).astype(F32)
    g = p.copy()
    g[y_gt] -= F32(1)
    dE = (wts.Wp.T @ g).astype(F32)
    return p, F32(-np.log(p[y_gt])), dE


def head_graph(U, wts, y_gt, Xin=None):
    """Graph mode: per-layer column-wise max over ALL rows (padded rows included)."""
    n = U[0].shape[0]
    acts = [np.maximum(U[0], 0), np.maximum(U[1], 0), U[2]] if Xin is None else [Xin[1], Xin[2], U[2]]
    arg = [a.argmax(0) for a in acts]
    e = np.concatenate([a.max(0) for a in acts]).astype(F32)
    z = (wts.Wp @ e + wts.bp).astype(F32)
    z = z - z.max()
    p = np.exp(z, dtype=F32)
    p = (p / p.sum(dtype=F32)).astype(F32)
    g = p.copy()
    g[y_gt] -= F32(1)
    dE = (wts.Wp.T @ g).astype(F32)
    return p, F32(-np.log(p[y_gt])), dE, arg


def direct_grads(dE, n, dims, rows):
    """Scatter dE (concat of 3 layer slices) to dense [n, d_l] 'direct' gradients.
    rows: node mode -> int t (all columns go to row t); graph mode -> list of 3 argmax index arrays."""
    out, off = [], 0
    for l, d in enumerate(dims):
        g = np.zeros((n, d), F32)
        if np.isscalar(rows) or isinstance(rows, (int, np.integer)):
            g[rows, :] = dE[off:off + d]
        else:
            g[rows[l], np.arange(d)] = dE[off:off + d]
        out.append(g)
        off += d
    return out


def backward(Abar, U, R, wts, dXd, bn=None):
    """dZ_l for l = 3, 2, 1 (as list index 2, 1, 0) and dX0 = Abar @ dZ_1.
    bn = (Xin, RS): batch-normalised activations and 1 / std of the two hidden layers."""
    dZ = [None, None, None]
    dX = dXd[2]
    for l in (2, 1, 0):
        if bn is not None and l < 2:          # backward of the row-wise standardisation: (dX - mean(dX) - x_hat mean(dX x_hat)) / std
            xh, rs = bn[0][l + 1], bn[1][l]
            dX = (rs * (dX - dX.mean(1, dtype=F32)[:, None] - xh * (dX * xh).mean(1, dtype=F32)[:, None])).astype(F32)
        dU = dX * (U[l] > 0) if l < 2 else dX
        dY = ((dU - U[l] * (dU * U[l]).sum(1, dtype=F32)[:, None]) / R[l][:, None]).astype(F32)
        dZ[l] = (dY @ wts.W[l].T).astype(F32)
        dX = (Abar @ dZ[l]).astype(F32)      # Abar is symmetric
        if l > 0:
            dX = dX + dXd[l - 1]
    return dZ, dX     # dX is dL/dX0


def grad_Abar(dZ, Xin, yhat, n, node_mode):
    """G = dL/dAbar = [dZ1|dZ2|dZ3] [X0|X1|X2]^T (+ Laplacian term in node mode)."""
    G = (np.concatenate(dZ, 1) @ np.concatenate(Xin, 1).T).astype(F32)
    if node_mode:
        y = yhat.astype(F32)
        G = G + ((y[None, :] ** 2 - y[:, None] * y[None, :]) / F32(n * n)).astype(F32)
    return G


def mask_grad(G, A, S, n):
    off = F32(1) - np.eye(n, dtype=F32)
    Gs = (G + G.T) / F32(2)
    ent = (np.log(F32(1) - S) - np.log(S)) / F32(n * n)
    return ((Gs * A * off + C_SIZE + C_ENT * ent) * S * (F32(1) - S)).astype(F32)


def adam(theta, m, v, g, step, lr):
    """torch.optim.Adam single-tensor update (fp32 state, scalars in double as torch does)."""
    m = (m + (g - m) * F32(1 - BETA1)).astype(F32)
    v = (v * F32(BETA2) + F32(1 - BETA2) * g * g).astype(F32)
    bc1 = 1.0 - BETA1 ** step
    bc2_sqrt = (1.0 - BETA2 ** step) ** 0.5
    denom = (np.sqrt(v) / F32(bc2_sqrt) + F32(EPS)).astype(F32)
    theta = (theta - F32(lr / bc1) * (m / denom)).astype(F32)
    return theta, m, v


class ClosedFormOracle:
    """Whole loop for one target; `stages` keeps the last iteration's intermediates."""

    def __init__(self, A, X, sd, gt_label, pred_label, node_idx, M0, graph_mode=False, lr=0.1, bn=False):
        self.bn = bool(bn)
        self.A = np.asarray(A, F32)
        self.X = np.asarray(X, F32)
        self.w = sd if isinstance(sd, Weights) else Weights(sd)
        self.n, self.D = self.X.shape
        self.t = int(node_idx)
        self.y_gt = int(gt_label)
        self.graph_mode = graph_mode
        self.yhat = None if graph_mode else np.asarray(pred_label, F32)
        self.M = np.asarray(M0, F32).copy()
        self.mM = np.zeros_like(self.M)
        self.vM = np.zeros_like(self.M)
        self.f = np.zeros(self.D, F32)
        self.mf = np.zeros(self.D, F32)
        self.vf = np.zeros(self.D, F32)
        self.lr = lr
        self.step = 0
        self.stages = {}
        self.trace = []
        self.trace_log = []

    def iterate(self):
        n, w = self.n, self.w
        Abar, S = masked_adj(self.M, self.A)
        phi = sigmoid(self.f)
        X0 = (self.X * phi).astype(F32)
        RS = None
        if self.bn:
            U, R, Xin, RS = forward(Abar, X0, w, bn=True)
        else:
            U, R, Xin = forward(Abar, X0, w)
        dims = [w.W[0].shape[1], w.W[1].shape[1], w.W[2].shape[1]]
        if self.graph_mode:
            p, pred_loss, dE, arg = head_graph(U, w, self.y_gt, Xin if self.bn else None)
            dXd = direct_grads(dE, n, dims, arg)
        else:
            p, pred_loss, dE = head_node(U, w, self.t, self.y_gt, Xin if self.bn else None)
            dXd = direct_grads(dE, n, dims, self.t)
        dZ, dX0 = backward(Abar, U, R, w, dXd, (Xin, RS) if self.bn else None)
        G = grad_Abar(dZ, Xin, self.yhat, n, not self.graph_mode)
        fsum = (dX0 * self.X).sum(0, dtype=F32)
        noise = getattr(self, "grad_noise", None)
        if noise is not None:      # conditioning probe (tests/golden/make_golden_noise_probe.py): every iteration, noise of +-1 ulp of the
            # MAGNITUDE OF THE SUMMANDS (sum of |products|: the forward error scale of a sum, whatever its order) on the two quantities
            # of the gradient that are long sums of products - dL/dAbar and the column sums of the feature-mask gradient.  Another
            # summation order, or an algebraically equal formula (the kernels form colsum(dZ1 (.) (Abar X)) where this oracle forms
            # colsum((Abar dZ1) (.) X)), differs by that much.  Where the summands cancel, or the prediction part of a gradient nearly
            # cancels its regulariser part, Adam's scale-free step turns it into a large relative change of the step - the amplification
            # that a perturbation of the START alone (make_golden_windows.py probe (ii)) under-samples.
            eps = F32(2.0 ** -23)
            Gabs = (np.abs(np.concatenate(dZ, 1)) @ np.abs(np.concatenate(Xin, 1)).T).astype(F32)
            G = (G + eps * Gabs * (2 * noise.random(G.shape, dtype=np.float32) - 1)).astype(F32)
            fabs = (np.abs(Abar) @ np.abs(dZ[0]) * np.abs(self.X)).sum(0, dtype=F32)
            fsum = (fsum + eps * fabs * (2 * noise.random(fsum.shape, dtype=np.float32) - 1)).astype(F32)
        dM = mask_grad(G, self.A, S, n)
        df = ((fsum + C_FEAT_SIZE / F32(self.D)) * phi * (F32(1) - phi)).astype(F32)
        # loss terms (logging parity only)
        size_l = C_SIZE * S.sum(dtype=F32)
        ent_l = C_ENT * (-S * np.log(S) - (F32(1) - S) * np.log(F32(1) - S)).mean(dtype=F32)
        fs_l = C_FEAT_SIZE * phi.mean(dtype=F32)
        if self.graph_mode:
            lap_l = F32(0)
        else:
            y = self.yhat
            lap_l = C_LAP * F32((y * y * Abar.sum(0, dtype=F32)).sum(dtype=F32) - y @ Abar @ y) / F32(n * n)
        loss = pred_loss + size_l + lap_l + ent_l + fs_l
        self.trace.append((float(loss), float(pred_loss), float(size_l), float(lap_l), float(ent_l), float(fs_l)))
        # what the reference prints next to the loss every epoch (explain.py:148-159): ExplainModule.mask_density (:680-683: sum of the masked
        # adjacency / sum of the adjacency) and the class probabilities of the forward (:710-714)
        self._log_p = np.asarray(p, F32).copy()
        self.stages = dict(Abar=Abar, S=S, phi=phi, X0=X0, U=U, R=R, Xin=Xin, p=p, dE=dE, dXd=dXd,
                           dZ=dZ, dX0=dX0, G=G, dM=dM, df=df)
        self.step += 1
        self.M, self.mM, self.vM = adam(self.M, self.mM, self.vM, dM, self.step, self.lr)
        self.f, self.mf, self.vf = adam(self.f, self.mf, self.vf, df, self.step, self.lr)
        # (the density is taken AFTER optimizer.step(), explain.py:142-148: the masked adjacency of the updated mask)
        with np.errstate(invalid="ignore", divide="ignore"):      # (a graph without edges: 0 / 0, as in the reference)
            self.trace_log.append((float(masked_adj(self.M, self.A)[0].sum(dtype=F32) / self.A.sum(dtype=F32)), self._log_p))

    def run(self, num_epochs):
        for _ in range(num_epochs):
            self.iterate()
        # explain.py:209-211: mask of the LAST forward (before the last step) times adj, float64
        return self.stages["Abar"].astype(np.float64) * self.A.astype(np.float64)
