"""
TEST DOUBLE for stochvolmodels_amd.engine.HipEngine: same methods, host memory, arithmetic by the CPU
oracle.  It exists so that the sharding / collective / step-offset logic of the chain drivers can run on a
machine without a GPU (world_size-2 gloo tests).  Never imported by the product.

"Pointers" are raw host addresses, handled exactly as the HIP engine handles device addresses.
"""
import ctypes as C

import numpy as np

from oracle import oracle


def _view(ptr, n):
    return np.ctypeslib.as_array((C.c_double * n).from_address(ptr))


class FakeEngine:
    torch_device = "cpu"
    device = 0

    def __init__(self, n_path, path_offset=0):
        self.n_path, self.path_offset = int(n_path), int(path_offset)
        self.x, self.vol, self.qvar = np.zeros(n_path), np.zeros(n_path), np.zeros(n_path)
        self._snap = None
        self._sums = {}
        self.calls = []

    def synchronize(self):
        pass

    def reserve_snapshots(self, rows):
        if self._snap is None or self._snap.shape[0] < rows:
            self._snap = np.zeros((rows, self.n_path))

    def snapshot_ptr(self, row):
        return self._snap[row].ctypes.data

    def snapshot(self, row, which="x"):
        self._snap[row] = self.x if which == "x" else self.qvar

    def alloc_sums(self, n, tag="sums"):
        buf = self._sums.get(tag)
        if buf is None or buf.size < n:
            buf = np.zeros(max(n, 1))
            self._sums[tag] = buf
        return buf.ctypes.data, buf

    def download(self, ptr, n):
        return _view(ptr, n).copy()

    def fill_state(self, x0, vol0, qvar0):
        self.x[:], self.vol[:], self.qvar[:] = x0, vol0, qvar0

    def get_state(self):
        return self.x.copy(), self.vol.copy(), self.qvar.copy()

    def logsv_rng(self, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, seed, call_id,
                  step_offset):
        self.calls.append(("logsv_rng", nb_steps, dt, step_offset))
        self.x, self.vol, self.qvar = oracle.logsv_terminal_rng(
            self.x, self.vol, self.qvar, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, seed, eta=eta,
            is_spot_measure=is_spot_measure, call_id=call_id, path_offset=self.path_offset, step_offset=step_offset)

    def heston_rng(self, nb_steps, dt, theta, kappa, rho, volvol, scheme, seed, call_id, step_offset):
        self.calls.append(("heston_rng", nb_steps, dt, step_offset))
        self.x, self.vol, self.qvar = oracle.heston_terminal_rng(
            self.x, self.vol, self.qvar, nb_steps, dt, theta, kappa, rho, volvol, seed, scheme=scheme,
            call_id=call_id, path_offset=self.path_offset, step_offset=step_offset)

    def finish_slice(self, forward, snap_row, qvar_row, spot_ptr):
        self.snapshot(snap_row, "x")
        if qvar_row is not None:
            self.snapshot(qvar_row, "qvar")
        self.spot_sums(self.snapshot_ptr(snap_row), forward, spot_ptr)

    def logsv_slice_rng(self, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, seed, call_id,
                        step_offset, forward, snap_row, qvar_row, spot_ptr, start=None):
        if start is not None:
            self.fill_state(*start)
        self.logsv_rng(nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, seed, call_id,
                       step_offset)
        self.finish_slice(forward, snap_row, qvar_row, spot_ptr)

    def logsv_chain_rng(self, nb_steps, dts, etas, forwards, theta, kappa1, kappa2, beta, volvol, is_spot_measure, seed,
                        call_id, step_offset, need_qvar, spot_ptr, start=None):
        if start is not None:
            self.fill_state(*start)
        m, step = len(nb_steps), step_offset
        for i in range(m):
            self.logsv_slice_rng(nb_steps[i], dts[i], theta, kappa1, kappa2, beta, volvol, float(etas[i]), is_spot_measure,
                                 seed, call_id, step, float(forwards[i]), i, (m + i) if need_qvar else None,
                                 spot_ptr + 16 * i)
            step += nb_steps[i]

    def heston_chain_rng(self, nb_steps, dts, forwards, theta, kappa, rho, volvol, scheme, seed, call_id, step_offset,
                         need_qvar, spot_ptr, start=None):
        if start is not None:
            self.fill_state(*start)
        m, step = len(nb_steps), step_offset
        for i in range(m):
            self.heston_slice_rng(nb_steps[i], dts[i], theta, kappa, rho, volvol, scheme, seed, call_id, step,
                                  float(forwards[i]), i, (m + i) if need_qvar else None, spot_ptr + 16 * i)
            step += nb_steps[i]

    def heston_slice_rng(self, nb_steps, dt, theta, kappa, rho, volvol, scheme, seed, call_id, step_offset, forward,
                         snap_row, qvar_row, spot_ptr, start=None):
        if start is not None:
            self.fill_state(*start)
        self.heston_rng(nb_steps, dt, theta, kappa, rho, volvol, scheme, seed, call_id, step_offset)
        self.finish_slice(forward, snap_row, qvar_row, spot_ptr)

    def upload_randoms(self, arrays, col0=0):
        self._rand = [np.ascontiguousarray(np.asarray(a)[:, col0:col0 + self.n_path]) for a in arrays]
        return tuple(range(len(arrays)))

    def logsv_w(self, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, w0, w1, ldw=None):
        self.x, self.vol, self.qvar = oracle.logsv_terminal_w(
            self.x, self.vol, self.qvar, dt, theta, kappa1, kappa2, beta, volvol, self._rand[w0], self._rand[w1],
            eta=eta, is_spot_measure=is_spot_measure)

    def rough_logsv(self, nb_steps, h, nodes, weights, v0, theta, kappa1, kappa2, rho, volvol, z0_ptr=None, z1_ptr=None,
                    ldw=None, seed=0, call_id=0, step_offset=0, from_origin=True, slice_out=None):
        assert from_origin
        if z0_ptr is None:
            Z0, Z1 = oracle.fill_normals(seed, self.n_path, nb_steps, call_id=call_id, path_offset=self.path_offset,
                                         step_offset=step_offset, stream=3)
        else:
            Z0, Z1 = self._rand[z0_ptr], self._rand[z1_ptr]
        n = len(nodes)
        ls, y = np.zeros(self.n_path), np.zeros(self.n_path)
        vol = np.ascontiguousarray(np.repeat(np.asarray(v0, dtype=float)[:, None], self.n_path, axis=1))
        L, p = oracle.lib(), oracle._p
        nodes, weights, v0 = (np.ascontiguousarray(a, dtype=np.float64) for a in (nodes, weights, v0))
        L.svo_rough_logsv_terminal_w(self.n_path, nb_steps, h, n, p(nodes), p(weights), p(v0), theta, kappa1, kappa2, rho,
                                     volvol, p(ls), p(vol), p(y), p(Z0), p(Z1), Z0.shape[1])
        self.x, self.qvar, self.factors = ls, y, vol
        if slice_out is not None:
            self.finish_slice(*slice_out)

    def rough_logsv_chain(self, nb_steps, hs, forwards, nodes, weights, v0, theta, kappa1, kappa2, rho, volvol, need_qvar,
                          spot_ptr, z0_ptr=None, z1_ptr=None, ldw=None, seed=0, call_id=0):
        m = len(nb_steps)
        for i in range(m):                                   # every expiry from time 0 on its own step
            self.rough_logsv(nb_steps[i], float(hs[i]), nodes, weights, v0, theta, kappa1, kappa2, rho, volvol, z0_ptr, z1_ptr,
                             ldw, seed, call_id, 0, True, (float(forwards[i]), i, (m + i) if need_qvar else None, spot_ptr + 16 * i))

    def logsv_slice_w(self, nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, w0, w1, forward,
                      snap_row, qvar_row, spot_ptr, ldw=None):
        self.logsv_w(nb_steps, dt, theta, kappa1, kappa2, beta, volvol, eta, is_spot_measure, w0, w1, ldw)
        self.finish_slice(forward, snap_row, qvar_row, spot_ptr)

    # the two reduction kernels, restated on host memory (utils/mc_payoffs.py:61-86)
    def spot_sums(self, x_ptr, forward, out_ptr):
        x = _view(x_ptr, self.n_path)
        spots = forward * np.exp(x)
        ok = ~np.isnan(spots)
        out = _view(out_ptr, 2)
        out[0], out[1] = spots[ok].sum(), ok.sum()

    def payoff_sums_chain(self, snap_rows, qvar_rows, forwards, ttms, spot_sums_ptr, strikes, codes, shifts, variable_type,
                          out_ptr):
        off = 0
        for i, (k, c, s) in enumerate(zip(strikes, codes, shifts)):
            self.payoff_sums(self.snapshot_ptr(list(snap_rows)[i]),
                             None if qvar_rows is None else self.snapshot_ptr(list(qvar_rows)[i]), float(forwards[i]),
                             float(ttms[i]), spot_sums_ptr + 16 * i, k, c, s, variable_type, out_ptr + 8 * off)
            off += 3 * len(k)

    def payoff_sums(self, x_ptr, qvar_ptr, forward, ttm, spot_sums_ptr, strikes, codes, shifts, variable_type, out_ptr):
        x = _view(x_ptr, self.n_path)
        ss = _view(spot_sums_ptr, 2)
        spots = forward * np.exp(x) - (ss[0] / ss[1] - forward)
        u = spots if variable_type == 1 else _view(qvar_ptr, self.n_path) / ttm
        out = _view(out_ptr, 3 * len(strikes))
        with np.errstate(all="ignore"):
            for k, (K, ty, sh) in enumerate(zip(strikes, codes, shifts)):
                pay = np.where(u > K, u - K, 0.0) if ty in (0, 2) else np.where(u < K, K - u, 0.0)
                if ty >= 2:
                    pay = pay / spots
                d = pay[~np.isnan(pay)] - sh
                out[3 * k:3 * k + 3] = d.sum(), (d * d).sum(), d.size
