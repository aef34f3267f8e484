"""CPU stand-in for ``smart_crossover.distributed.HipOps`` in the gloo rehearsals (TEST INFRASTRUCTURE): the same
methods, the rank-local kernels played by the oracle on torch CPU tensors.  What the rehearsals exercise is the
product's partitioning, protocol and collectives (smart_crossover/distributed.py); the kernels themselves are
compared with the oracle in the -m gpu tests."""
import numpy as np
import scipy.sparse as sp
import torch

from oracle import lp_path as L
from oracle import net_path as N


class _Mat:
    def __init__(self, csr):
        self.csr = sp.csr_matrix(csr)
        self.shape = self.csr.shape


class OracleOps:
    def vec(self, host):
        return torch.from_numpy(np.ascontiguousarray(host).copy())

    def empty(self, n, dtype):
        return torch.zeros(int(n), dtype={np.float64: torch.float64, np.uint8: torch.uint8, np.int64: torch.int64}[dtype])

    def host(self, t):
        return t.numpy()

    def matrix(self, csr):
        return _Mat(csr)

    row_matrix = matrix

    # ---- LP
    def score_columns(self, A, y, c, x, l, u, gamma, code):
        s_d = L.dual_slack(A.csr, c.numpy(), y.numpy())
        code.copy_(torch.from_numpy(L.column_codes(x.numpy(), l.numpy(), u.numpy(), s_d, gamma)))

    def score_rows(self, A_rows, x, b, y, gamma_dual, flag):
        s_p = L.primal_slack(A_rows.csr, b.numpy(), x.numpy())
        flag.copy_(torch.from_numpy(L.row_flags(s_p, y.numpy(), gamma_dual).astype(np.uint8)))

    def count(self, flags, mask):
        return int(np.count_nonzero(flags.numpy() & mask))

    def price(self, A, y, c, vbasis, tol):
        rc = L.dual_slack(A.csr, c.numpy(), y.numpy())
        if vbasis is not None:
            rc = np.where(vbasis.numpy() == -2, -rc, rc)
        if rc.size == 0:
            return (float("nan"), -1, 0)
        j = int(np.argmin(rc))
        return (float(rc[j]), j, int(np.count_nonzero(~(rc >= -tol))))

    def dual_slack(self, A, y, c):
        return torch.from_numpy(L.dual_slack(A.csr, c.numpy(), y.numpy()))

    def fixed_rhs(self, A_rows, code_all, u_all, l_all, b_loc, out):
        code = code_all.numpy()
        up, low = np.flatnonzero(code & 2), np.flatnonzero((code & 1) & ~((code & 2) >> 1))
        A = A_rows.csr
        out.copy_(torch.from_numpy(b_loc.numpy() - A[:, up] @ u_all.numpy()[up] - A[:, low] @ l_all.numpy()[low]))

    # ---- sharded CG: the matrix-free recurrence of oracle.lp_path.cg_legacy, one step at a time
    def cg_open(self, A, xa, xs, c, tol):
        xa_, xs_, c_ = xa.numpy(), xs.numpy(), c.numpy()
        q = torch.from_numpy(A.csr @ (xa_ * xa_ * c_))
        return {"A": A.csr, "xa": xa_, "xs": xs_, "c": c_, "tol": tol, "q": q, "iters": 0, "done": False, "conv": False}

    def cg_start(self, s):
        b = s["q"].numpy().copy()
        s["r"], s["p"], s["z"] = b.copy(), b.copy(), np.zeros_like(b)
        s["rho"] = float(b @ b)
        s["bn"] = float(np.sqrt(s["rho"]))
        s["atol"] = s["tol"] * s["bn"]
        trivial = not (s["bn"] > s["tol"]) or s["bn"] < s["atol"]
        s["done"] = trivial
        return s["bn"], trivial

    def cg_local(self, s):
        if s["done"]:
            return
        w = s["xa"] ** 2 * (s["A"].T @ s["p"])
        s["q"].copy_(torch.from_numpy(s["A"] @ w))

    def cg_update(self, s, k):
        if s["done"]:
            return
        q = s["q"].numpy() + s["xs"] ** 2 * s["p"]
        alpha = s["rho"] / float(s["p"] @ q)
        s["z"] = s["z"] + alpha * s["p"]
        s["r"] = s["r"] - alpha * q
        rho_new = float(s["r"] @ s["r"])
        s["iters"] += 1
        if np.sqrt(rho_new) < s["atol"]:
            s["done"] = s["conv"] = True
        else:
            s["p"] = s["r"] + (rho_new / s["rho"]) * s["p"]
        s["rho"] = rho_new

    def cg_poll(self, s):
        return s["done"], s["iters"]

    def cg_finish(self, s):
        cols = s["xa"] * (s["c"] - s["A"].T @ s["z"])
        rows = -(s["xs"] * s["z"])
        return float(cols @ cols), float(rows @ rows), s["iters"], s["conv"]

    # ---- MCF (oracle.net_path, step by step)
    def mcf_xhat(self, x, u, xhat, mask):
        xh, big = N.mcf_signed_residual_flow(x.numpy(), u.numpy())
        xhat.copy_(torch.from_numpy(xh))
        mask.copy_(torch.from_numpy(big.astype(np.uint8)))

    def mcf_node_flows(self, A_rows, xhat_all, mask_all, f_inv):
        f1, f2 = N.mcf_node_throughput(A_rows.csr, xhat_all.numpy(), mask_all.numpy().astype(bool))
        f = np.maximum(f1, f2)
        out = np.zeros_like(f)
        nz = f != 0
        out[nz] = 1 / f[nz]
        f_inv.copy_(torch.from_numpy(out))

    def mcf_arc_indicator(self, A_cols, xhat, mask, f_inv_all, ind):
        C = sp.csc_matrix(A_cols.csr)
        C.sum_duplicates()
        n = C.shape[1]
        cols = np.repeat(np.arange(n), np.diff(C.indptr))
        abar = C.data * np.where(mask.numpy().astype(bool), -1.0, 1.0)[cols]
        r = np.abs((f_inv_all.numpy()[C.indices] * xhat.numpy()[cols]) * abar)
        keep = abar != 0
        out = np.zeros(n)
        np.maximum.at(out, cols[keep], r[keep])
        ind.copy_(torch.from_numpy(out))

    def top_k(self, key, k):
        keys = key.numpy()
        order = np.argsort(keys, kind="stable")[::-1][:int(k)]
        return keys[order], order
