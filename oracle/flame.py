"""CPU restatement of the FLAME vertex path (TEST INFRASTRUCTURE ONLY, see oracle/__init__.py).

Follows third_party/inferno/inferno/utils/lbs.py (``lbs`` :142-235, ``vertices2joints`` :260-277,
``blend_shapes`` :280-301, ``batch_rodrigues`` :304-335, ``transform_mat`` :338-348, ``batch_rigid_transform``
:351-410) and ``FLAME.forward`` in third_party/inferno/inferno/models/DecaFLAME.py:222-270 (pose assembly
[global | neck | jaw | eyes], betas = [shape | expression]); landmarks (:246-268) are not part of this path.
Pinned by tests/golden/flame.npz, which is produced by IMPORTING the reference's own lbs.py on a synthetic basis
(avi_talking_amd.weights.make_flame_basis: the licensed generic_model.pkl is absent)."""
import torch


def batch_rodrigues(rot_vecs):
    """lbs.py:304-335 (note the +1e-8 inside the norm, kept bit for bit)."""
    n = rot_vecs.shape[0]
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    rot_dir = rot_vecs / angle
    cos, sin = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    rx, ry, rz = torch.split(rot_dir, 1, dim=1)
    z = torch.zeros((n, 1), dtype=rot_vecs.dtype)
    K = torch.cat([z, -rz, ry, rz, z, -rx, -ry, rx, z], dim=1).view(n, 3, 3)
    ident = torch.eye(3, dtype=rot_vecs.dtype)[None]
    return ident + sin * K + (1 - cos) * torch.bmm(K, K)


def batch_rigid_transform(rot_mats, joints, parents):
    """lbs.py:351-410: chain of 4x4 transforms, made relative to the rest-pose joints."""
    B, J = joints.shape[:2]
    joints = joints[..., None]
    rel = joints.clone()
    rel[:, 1:] -= joints[:, parents[1:]]
    T = torch.zeros(B, J, 4, 4, dtype=joints.dtype)
    T[:, :, :3, :3] = rot_mats
    T[:, :, :3, 3:] = rel
    T[:, :, 3, 3] = 1.0
    chain = [T[:, 0]]
    for i in range(1, J):
        chain.append(torch.matmul(chain[int(parents[i])], T[:, i]))
    tr = torch.stack(chain, dim=1)
    posed = tr[:, :, :3, 3]
    jh = torch.nn.functional.pad(joints, [0, 0, 0, 1])
    rel_tr = tr - torch.nn.functional.pad(torch.matmul(tr, jh), [3, 0, 0, 0, 0, 0, 0, 0])
    return posed, rel_tr


def lbs(betas, pose, basis):
    """lbs.py:142-235 with pose2rot=True.  betas (N, n_shape+n_exp), pose (N, 15) axis-angle -> (N, V, 3), (N, 5, 3)."""
    N = betas.shape[0]
    v_shaped = basis["v_template"][None] + torch.einsum("bl,mkl->bmk", betas, basis["shapedirs"])
    J = torch.einsum("bik,ji->bjk", v_shaped, basis["J_regressor"])
    rot = batch_rodrigues(pose.reshape(-1, 3)).view(N, -1, 3, 3)
    feat = (rot[:, 1:] - torch.eye(3)).reshape(N, -1)
    v_posed = v_shaped + torch.matmul(feat, basis["posedirs"]).view(N, -1, 3)
    Jt, A = batch_rigid_transform(rot, J, basis["parents"])
    Tm = torch.matmul(basis["lbs_weights"][None].expand(N, -1, -1), A.view(N, 5, 16)).view(N, -1, 4, 4)
    vh = torch.cat([v_posed, torch.ones(N, v_posed.shape[1], 1)], dim=2)
    return torch.matmul(Tm, vh[..., None])[:, :, :3, 0], Jt


def flame_forward(basis, shape_params, expression_params, pose_params, eye_pose_params=None, neck_pose=None):
    """DecaFLAME.py:222-244: vertices of FLAME.forward (pose_params = [global(3) | jaw(3)])."""
    N = shape_params.shape[0]
    eye = torch.zeros(N, 6) if eye_pose_params is None else eye_pose_params
    neck = torch.zeros(N, 3) if neck_pose is None else neck_pose
    betas = torch.cat([shape_params, expression_params], dim=1)
    full_pose = torch.cat([pose_params[:, :3], neck, pose_params[:, 3:], eye], dim=1)
    return lbs(betas, full_pose, basis)[0]
