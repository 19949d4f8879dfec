"""Camera-pose utilities on the host, either side of the render path: what turns an LLFF / Objectron
``poses_bounds.npy`` into the poses render() is called with and into the spiral or circle of poses render_path() is
driven by (nerf-pytorch/load_llff.py:58-110, 126-316), the Blender-style pose on a sphere (load_blender.py:15-34),
and a dependency-free PNG writer for render_path's frame dumps (run_nerf.py:170-175 uses imageio, which is optional
here).  Pure numpy: this is host set-up work, done once per scene.
"""
import struct
import zlib

import numpy as np


# ---------------------------------------------------------------------------- small vector helpers (load_llff.py:126-150)
def normalize(x):
    return x / np.linalg.norm(x)


def viewmatrix(z, up, pos):
    """3x4 camera-to-world frame looking along z with the given up hint, at pos."""
    back = normalize(z)
    right = normalize(np.cross(up, back))
    true_up = normalize(np.cross(back, right))
    return np.stack([right, true_up, back, pos], 1)


def ptstocam(pts, c2w):
    return np.matmul(c2w[:3, :3].T, (pts - c2w[:3, 3])[..., np.newaxis])[..., 0]


def poses_avg(poses):
    """Average pose [3,5] of poses [N,3,5]: mean position, summed viewing and up axes; the hwf column of the first."""
    hwf = poses[0, :3, -1:]
    center = poses[:, :3, 3].mean(0)
    back = normalize(poses[:, :3, 2].sum(0))
    up = poses[:, :3, 1].sum(0)
    return np.concatenate([viewmatrix(back, up, center), hwf], 1)


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """N poses [3,5] on a spiral around c2w, all looking at the point `focal` in front of it (load_llff.py:155-164)."""
    out = []
    rads = np.array(list(rads) + [1.])
    hwf = c2w[:, 4:5]
    for theta in np.linspace(0., 2. * np.pi * rots, int(N) + 1)[:-1]:
        offset = np.array([np.cos(theta), -np.sin(theta), -np.sin(theta * zrate), 1.]) * rads
        eye = np.dot(c2w[:3, :4], offset)
        target = np.dot(c2w[:3, :4], np.array([0, 0, -focal, 1.]))
        out.append(np.concatenate([viewmatrix(normalize(eye - target), up, eye), hwf], 1))
    return out


def _to44(p):
    last = np.tile(np.reshape(np.eye(4)[-1, :], [1, 1, 4]), [p.shape[0], 1, 1])
    return np.concatenate([p, last], 1)


def recenter_poses(poses):
    """Expresses all poses in the frame of their average (load_llff.py:168-181); the hwf column is kept."""
    out = poses + 0
    avg = poses_avg(poses)
    avg44 = np.concatenate([avg[:3, :4], np.reshape([0, 0, 0, 1.], [1, 4])], -2)
    out[:, :3, :4] = (np.linalg.inv(avg44) @ _to44(poses[:, :3, :4]))[:, :3, :4]
    return out


def spherify_poses(poses, bds):
    """Inward-facing captures: recentre on the point closest to all optical axes, scale the cameras onto the unit
    sphere, and return a 120-pose circle at their mean height (load_llff.py:188-245).  bds is scaled IN PLACE, like
    the reference does.  Returns (poses_reset [N,3,5], circle [120,3,5], bds)."""
    dirs = poses[:, :3, 2:3]
    origins = poses[:, :3, 3:4]
    A = np.eye(3) - dirs * np.transpose(dirs, [0, 2, 1])
    b = -A @ origins
    center = np.squeeze(-np.linalg.inv((np.transpose(A, [0, 2, 1]) @ A).mean(0)) @ b.mean(0))
    up = (poses[:, :3, 3] - center).mean(0)
    e0 = normalize(up)
    e1 = normalize(np.cross([.1, .2, .3], e0))
    e2 = normalize(np.cross(e0, e1))
    frame = np.stack([e1, e2, e0, center], 1)
    reset = np.linalg.inv(_to44(frame[None])) @ _to44(poses[:, :3, :4])
    rad = np.sqrt(np.mean(np.sum(np.square(reset[:, :3, 3]), -1)))
    sc = 1. / rad
    reset[:, :3, 3] *= sc
    bds *= sc
    rad *= sc
    zh = np.mean(reset[:, :3, 3], 0)[2]
    radcircle = np.sqrt(rad ** 2 - zh ** 2)
    circle = []
    for th in np.linspace(0., 2. * np.pi, 120):
        eye = np.array([radcircle * np.cos(th), radcircle * np.sin(th), zh])
        back = normalize(eye)
        right = normalize(np.cross(back, np.array([0, 0, -1.])))
        down = normalize(np.cross(back, right))
        circle.append(np.stack([right, down, back, eye], 1))
    circle = np.stack(circle, 0)
    hwf = poses[0, :3, -1:]
    circle = np.concatenate([circle, np.broadcast_to(hwf, circle[:, :3, -1:].shape)], -1)
    reset = np.concatenate([reset[:, :3, :4], np.broadcast_to(hwf, reset[:, :3, -1:].shape)], -1)
    return reset, circle, bds


# ---------------------------------------------------------------------------- poses_bounds.npy (load_llff.py:58-110, 247-316)
def parse_poses_bounds(poses_arr, image_hw, pose_factor=1):
    """poses_bounds.npy [N,17] -> (poses [3,5,N], bds [2,N]) as _load_data leaves them: the hwf column gets the size
    of the images actually used and the focal length divided by pose_factor (load_llff.py:62-64, 103-105)."""
    poses_arr = np.asarray(poses_arr)
    poses = poses_arr[:, :-2].reshape([-1, 3, 5]).transpose([1, 2, 0]).copy()
    bds = poses_arr[:, -2:].transpose([1, 0]).copy()
    poses[:2, 4, :] = np.array(image_hw[:2]).reshape([2, 1])
    poses[2, 4, :] = poses[2, 4, :] * 1. / pose_factor
    return poses, bds


def llff_poses(poses, bds, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """The pose half of load_llff_data (load_llff.py:251-316): axis reordering from the LLFF convention, rescaling by
    the near bound, recentring, and the render path (spiral, or circle when spherify).  poses [3,5,N], bds [2,N] as
    parse_poses_bounds returns them.  Returns (poses [N,3,5] float32, bds [N,2] float32, render_poses [M,3,5]
    float32, i_test)."""
    poses = np.concatenate([poses[:, 1:2, :], -poses[:, 0:1, :], poses[:, 2:, :]], 1)
    poses = np.moveaxis(poses, -1, 0).astype(np.float32)
    bds = np.moveaxis(bds, -1, 0).astype(np.float32)
    sc = 1. if bd_factor is None else 1. / (bds.min() * bd_factor)
    poses[:, :3, 3] *= sc
    bds *= sc
    if recenter:
        poses = recenter_poses(poses)
    if spherify:
        poses, render_poses, bds = spherify_poses(poses, bds)
    else:
        c2w = poses_avg(poses)
        up = normalize(poses[:, :3, 1].sum(0))
        close_depth, inf_depth = bds.min() * .9, bds.max() * 5.
        dt = .75
        focal = 1. / (((1. - dt) / close_depth + dt / inf_depth))
        zdelta = close_depth * .2
        rads = np.percentile(np.abs(poses[:, :3, 3]), 90, 0)
        n_views, n_rots = 120, 2
        if path_zflat:
            zloc = -close_depth * .1
            c2w[:3, 3] = c2w[:3, 3] + zloc * c2w[:3, 2]
            rads[2] = 0.
            n_rots = 1
            n_views /= 2
        render_poses = render_path_spiral(c2w, up, rads, focal, zdelta, zrate=.5, rots=n_rots, N=n_views)
    render_poses = np.array(render_poses).astype(np.float32)
    c2w = poses_avg(poses)
    i_test = int(np.argmin(np.sum(np.square(c2w[:3, 3] - poses[:, :3, 3]), -1)))
    return poses.astype(np.float32), bds, render_poses, i_test


# ---------------------------------------------------------------------------- load_blender.py:15-34
def pose_spherical(theta, phi, radius):
    """4x4 camera-to-world on a sphere around the origin (degrees), as a float32 torch tensor like the reference's."""
    import torch
    from .synth import pose_spherical as _ps
    return torch.from_numpy(_ps(theta, phi, radius))


# ---------------------------------------------------------------------------- PNG (8-bit RGB / grey), no dependencies
def write_png(path, img8):
    """img8: uint8 [H,W,3] or [H,W].  Plain zlib-compressed, filter 0 scanlines."""
    img8 = np.ascontiguousarray(img8)
    if img8.dtype != np.uint8 or img8.ndim not in (2, 3) or (img8.ndim == 3 and img8.shape[2] not in (1, 3, 4)):
        raise ValueError("write_png expects uint8 [H,W], [H,W,3] or [H,W,4], got %s %s" % (img8.dtype, img8.shape))
    h, w = img8.shape[:2]
    ch = 1 if img8.ndim == 2 else img8.shape[2]
    color = {1: 0, 3: 2, 4: 6}[ch]
    rows = np.concatenate([np.zeros((h, 1), np.uint8), img8.reshape(h, w * ch)], 1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, color, 0, 0, 0)) +
                chunk(b"IDAT", zlib.compress(rows, 6)) + chunk(b"IEND", b""))


def read_png(path):
    """Reads back what write_png wrote (8-bit, non-interlaced, any scanline filter)."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:8] != b"\x89PNG\r\n\x1a\n":
        raise ValueError("%s is not a PNG file" % path)
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n, tag = struct.unpack(">I", data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        body = data[pos + 8:pos + 8 + n]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, depth, color, _, _, interlace = hdr
    if depth != 8 or interlace != 0 or color not in (0, 2, 6):
        raise ValueError("read_png handles 8-bit non-interlaced grey / RGB / RGBA only")
    ch = {0: 1, 2: 3, 6: 4}[color]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch)
    out = np.zeros((h, w * ch), np.uint8)
    for y in range(h):
        ft, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        prev = out[y - 1].astype(np.int32) if y else np.zeros(w * ch, np.int32)
        if ft == 0:
            rec = line
        elif ft == 2:
            rec = (line + prev) & 255
        else:
            rec = np.zeros(w * ch, np.int32)
            for x in range(w * ch):
                a = rec[x - ch] if x >= ch else 0
                b = prev[x]
                c = prev[x - ch] if x >= ch else 0
                if ft == 1:
                    p = a
                elif ft == 3:
                    p = (a + b) // 2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    p = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                rec[x] = (line[x] + p) & 255
        out[y] = rec
    return out.reshape(h, w) if ch == 1 else out.reshape(h, w, ch)
