"""Host-side mirror of the FLAME vertex path: ``inferno/models/DecaFLAME.py`` ``FLAME.forward`` (vertices only) and
the call pattern of ``FlamePreprocessor._forward`` (inferno/models/temporal/Preprocessors.py:62-160: per-frame
expression + jaw, zero global rotation, one shape vector per clip).

The FLAME buffers (``v_template``, ``shapedirs``, ``posedirs``, ``J_regressor``, ``lbs_weights``; same names as
DecaFLAME.py:60-85) are re-laid out once; the arithmetic runs in ``avi_flame_vertices`` (csrc/flame.hip)."""
import ctypes as C

import torch

from .. import lib as L


class FLAME:
    def __init__(self, buffers, n_shape=300, n_exp=50, device="cuda", matrix_cores=True):
        """``buffers``: dict with the FLAME buffer names; ``shapedirs`` is (V, 3, n_shape + n_exp) as registered by
        the reference after its slicing (DecaFLAME.py:65-67)."""
        self.device = torch.device(device)
        self.n_shape, self.n_exp = n_shape, n_exp
        f64 = lambda k: buffers[k].detach().to(torch.float64)
        vt, sd, pd, jr = f64("v_template"), f64("shapedirs"), f64("posedirs"), f64("J_regressor")
        V = vt.shape[0]
        if sd.shape != (V, 3, n_shape + n_exp) or pd.shape != (36, V * 3) or jr.shape != (5, V):
            raise ValueError("unexpected FLAME buffer shapes")
        parents = buffers.get("parents")
        if parents is not None and parents.tolist() != [-1, 0, 1, 1, 1]:
            raise ValueError("the kernel hard-codes FLAME's kinematic tree [-1, 0, 1, 1, 1]")
        basis = sd.reshape(V * 3, -1).t().contiguous()                      # [n_shape + n_exp][V*3]
        dev = lambda t: t.to(self.device, torch.float32).contiguous()
        self._keep = {
            "v_template": dev(vt.reshape(-1)),
            "shape_basis": dev(basis[:n_shape]),
            "frame_basis": dev(torch.cat([basis[n_shape:], pd], 0)),        # [n_exp + 36][V*3]
            "j_template": dev(jr @ vt),                                     # folded in float64 once
            "j_shape": dev(torch.einsum("jv,vck->jck", jr, sd[..., :n_shape]).reshape(15, n_shape)),
            "j_exp": dev(torch.einsum("jv,vck->jck", jr, sd[..., n_shape:]).reshape(15, n_exp)),
            "lbs_weights": dev(f64("lbs_weights")),
        }
        fb = L.AviFlameBasis()
        for k, t in self._keep.items():
            setattr(fb, k, t.data_ptr())
        fb.V, fb.n_shape, fb.n_exp = V, n_shape, n_exp
        fb.basis_hi = fb.basis_lo = None
        # matrix-core kernel: up to 160 per-frame basis vectors (n_exp <= 124; FLAME has 100, the reference uses 50);
        # the fp32 vector-pipe kernel keeps all of them in LDS: up to 106 (n_exp <= 70)
        if n_exp + 36 > (160 if matrix_cores else 106):
            raise ValueError(f"n_exp = {n_exp}: more per-frame basis vectors than the "
                             f"{'matrix-core' if matrix_cores else 'vector-pipe'} kernel holds")
        if matrix_cores:
            # split bf16 planes of the per-frame basis, [3][Vp][96|160]: the blend runs on the matrix cores
            KP, Vp = (96 if n_exp + 36 <= 96 else 160), (V + 15) // 16 * 16
            self._planes = torch.empty((2, 3 * Vp * KP), dtype=torch.int16, device=self.device)
            L.check(L.load().avi_flame_pack_basis(C.byref(fb), self._planes[0].data_ptr(), self._planes[1].data_ptr(),
                                                  L.stream_ptr()), "avi_flame_pack_basis")
            fb.basis_hi, fb.basis_lo = self._planes[0].data_ptr(), self._planes[1].data_ptr()
        self.fb, self.V = fb, V

    def vertices(self, shape, exp, pose15):
        """shape (B, n_shape) per clip, exp (B, T, n_exp), pose15 (B, T, 15) full axis-angle pose -> (B, T, V, 3)."""
        L.require_gpu(shape, exp, pose15)
        B, T = exp.shape[:2]
        f = lambda t: t.to(self.device, torch.float32).contiguous()
        shape, exp, pose15 = f(shape), f(exp), f(pose15)
        F, K = B * T, self.n_exp + 36
        vsh = torch.empty((B, self.V * 3), dtype=torch.float32, device=self.device)
        tiles = B * ((T + 15) // 16)                      # matrix-core path: operands in 16-frame fragment tiles
        coef = torch.empty((max((F + 7) // 8 * 8 * max(K, 160), tiles * 2560),), dtype=torch.float32, device=self.device)
        xf = torch.empty((max(F * 60, tiles * 3072) + B * 16,), dtype=torch.float32, device=self.device)
        out = torch.empty((B, T, self.V, 3), dtype=torch.float32, device=self.device)
        L.check(L.load().avi_flame_vertices(C.byref(self.fb), shape.data_ptr(), exp.data_ptr(), pose15.data_ptr(), B, T,
                                            vsh.data_ptr(), coef.data_ptr(), xf.data_ptr(), out.data_ptr(),
                                            L.stream_ptr()), "avi_flame_vertices")
        return out

    def forward(self, shape_params=None, expression_params=None, pose_params=None, eye_pose_params=None,
                neck_pose=None):
        """DecaFLAME.py:222-244 for N independent frames (each with its own shape): returns the vertices (N, V, 3)
        (the reference also returns two landmark sets, which are not part of this path: ``None`` here)."""
        N = shape_params.shape[0]
        dev = self.device
        z = lambda n: torch.zeros((N, n), dtype=torch.float32, device=dev)
        exp = z(self.n_exp) if expression_params is None else expression_params
        pose = z(6) if pose_params is None else pose_params
        eye = z(6) if eye_pose_params is None else eye_pose_params
        neck = z(3) if neck_pose is None else neck_pose
        full = torch.cat([pose[:, :3], neck, pose[:, 3:], eye], dim=1).to(dev, torch.float32)   # DecaFLAME.py:240
        v = self.vertices(shape_params, exp.reshape(N, 1, -1), full.reshape(N, 1, 15))
        return v.reshape(N, self.V, 3), None, None

    __call__ = forward

    def from_coefficients(self, gt_shape, predicted_exp, predicted_jaw):
        """``FlamePreprocessor._forward`` for one reconstruction type: shape (B, n_shape), exp (B, T, >= n_exp),
        jaw (B, T, 3) -> vertices (B, T, V*3) (global rotation zero, Preprocessors.py:88-90)."""
        B, T = predicted_exp.shape[:2]
        pose = torch.zeros((B, T, 15), dtype=torch.float32, device=self.device)
        pose[..., 6:9] = predicted_jaw
        return self.vertices(gt_shape[:, : self.n_shape], predicted_exp[..., : self.n_exp], pose).reshape(B, T, -1)
