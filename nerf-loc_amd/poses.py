"""Camera-pose utilities on the host, either side of the render path: what turns an LLFF / Objectron
``poses_bounds.npy`` into the poses render() is called with and into the spiral or circle of poses render_path() is
driven by (nerf-pytorch/load_llff.py:58-110, 126-316), the Blender-style pose on a sphere (load_blender.py:15-34),
and dependency-free writers for render_path's outputs: PNG frame dumps (run_nerf.py:170-175) and AVI videos
(run_nerf.py:681, 819-820 write MP4); the reference uses imageio for both, which is optional here.  Pure numpy: this is
host set-up work, done once per scene.
"""
import struct
import zlib

import numpy as np


# ---------------------------------------------------------------------------- frames
# A camera is a 3x4 frame [right | up | back | eye] (columns; the camera looks along -back), optionally with a fifth
# column (height, width, focal).  Everything below works on stacks of frames at once; rigid frames are inverted as
# [R^T | -R^T t], not through a general 4x4 inverse.
def _unit(v):
    v = np.asarray(v, dtype=np.float64)
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def _frames(back, up_hint, eye):
    """Stack of frames [..., 3, 4] with the given viewing axes `back` [..., 3] (need not be unit), up hints and eyes."""
    back = _unit(back)
    right = _unit(np.cross(np.broadcast_to(up_hint, back.shape), back))
    up = np.cross(back, right)                     # unit already: back and right are orthonormal
    return np.stack([right, up, back, np.broadcast_to(np.asarray(eye, np.float64), back.shape)], -1)


def _into_frame(frame, cams):
    """Frames `cams` [N, 3, 4] expressed in the coordinates of the rigid `frame` [3, 4]."""
    R, t = frame[:, :3], frame[:, 3]
    out = np.empty(cams.shape, np.float64)
    out[:, :, :3] = np.einsum('ji,njk->nik', R, cams[:, :, :3])
    out[:, :, 3] = (cams[:, :, 3] - t) @ R
    return out


def _with_hwf(frames, hwf):
    return np.concatenate([frames, np.broadcast_to(np.asarray(hwf, np.float64).reshape(3, 1), frames.shape[:-1] + (1,))], -1)


def normalize(x):
    """x / |x| (load_llff.py:126-127)."""
    return _unit(x)


def viewmatrix(z, up, pos):
    """3x4 camera-to-world frame looking along z with the given up hint, at pos (load_llff.py:129-136)."""
    return _frames(np.asarray(z, np.float64), np.asarray(up, np.float64), pos)


def ptstocam(pts, c2w):
    """World points [..., 3] in the camera's coordinates (load_llff.py:138-140)."""
    c2w = np.asarray(c2w, np.float64)
    return (np.asarray(pts, np.float64) - c2w[:3, 3]) @ c2w[:3, :3]


def poses_avg(poses):
    """Average camera [3,5] of poses [N,3,5] (load_llff.py:142-151): the mean eye, the frame spanned by the summed
    viewing axes with the summed up axes as hint, and the (height, width, focal) column of the first pose."""
    poses = np.asarray(poses)
    frame = _frames(poses[:, :3, 2].sum(0, dtype=np.float64), poses[:, :3, 1].sum(0, dtype=np.float64),
                    poses[:, :3, 3].mean(0, dtype=np.float64))
    return _with_hwf(frame, poses[0, :3, 4])


def render_path_spiral(c2w, up, rads, focal, zdelta, zrate, rots, N):
    """N poses [3,5] on a spiral around the camera c2w, every one looking at the point `focal` in front of it
    (load_llff.py:155-164): in camera coordinates the eye runs over (cos t, -sin t, -sin(zrate t)) * rads for t in
    [0, 2 pi rots).  zdelta is accepted and unused, as in the reference.  Returns a list of N arrays."""
    c2w = np.asarray(c2w, np.float64)
    t = np.linspace(0., 2. * np.pi * rots, int(N) + 1)[:-1]
    local = np.stack([np.cos(t), -np.sin(t), -np.sin(t * zrate)], -1) * np.asarray(rads, np.float64)[:3]
    eyes = local @ c2w[:3, :3].T + c2w[:3, 3]
    target = c2w[:3, 3] - focal * c2w[:3, 2]
    return list(_with_hwf(_frames(eyes - target, np.asarray(up, np.float64), eyes), c2w[:3, 4]))


def recenter_poses(poses):
    """All poses [N,3,5] in the coordinates of their average camera (load_llff.py:168-186); the hwf column is kept."""
    poses = np.asarray(poses)
    out = np.array(poses, dtype=np.float64)
    out[:, :3, :4] = _into_frame(poses_avg(poses)[:3, :4], out[:, :3, :4])
    return out.astype(poses.dtype)


def spherify_poses(poses, bds):
    """Inward-facing captures (load_llff.py:189-245): origin at the point closest, in the least-squares sense, to all
    optical axes; z along the mean offset of the cameras from it; cameras scaled so that their root-mean-square
    distance is 1; plus a circle of 120 cameras at their mean height looking at the origin.  bds is scaled IN PLACE,
    like the reference does.  Returns (poses [N,3,5], circle [120,3,5], bds)."""
    poses = np.asarray(poses)
    cams = np.array(poses[:, :3, :4], dtype=np.float64)
    d, o = cams[:, :, 2], cams[:, :, 3]
    # sum_i |P_i (c - o_i)|^2 -> min with P_i = I - d_i d_i^T: normal equations (mean P_i^T P_i) c = mean P_i^T P_i o_i
    P = np.eye(3) - d[:, :, None] * d[:, None, :]
    center = np.linalg.solve(np.einsum('nji,njk->ik', P, P) / len(P), np.einsum('nij,nj->i', P, o) / len(P))
    e_z = _unit((o - center).mean(0))
    e_x = _unit(np.cross([.1, .2, .3], e_z))       # the reference's helper direction fixes the in-plane orientation
    e_y = np.cross(e_z, e_x)
    cams = _into_frame(np.stack([e_x, e_y, e_z, center], -1), cams)
    scale = 1. / np.sqrt(np.mean(np.sum(cams[:, :, 3] ** 2, -1)))
    cams[:, :, 3] *= scale
    bds *= scale
    height = cams[:, 2, 3].mean()
    ring = np.sqrt(1. - height ** 2)               # the scaled cameras have unit root-mean-square distance
    th = np.linspace(0., 2. * np.pi, 120)
    eyes = np.stack([ring * np.cos(th), ring * np.sin(th), np.full_like(th, height)], -1)
    back = _unit(eyes)
    right = _unit(np.cross(back, [0., 0., -1.]))
    circle = np.stack([right, _unit(np.cross(back, right)), back, eyes], -1)
    hwf = poses[0, :3, 4]
    return _with_hwf(cams, hwf), _with_hwf(circle, hwf), bds


# ---------------------------------------------------------------------------- poses_bounds.npy (load_llff.py:58-110, 247-316)
def parse_poses_bounds(poses_arr, image_hw, pose_factor=1):
    """poses_bounds.npy [N,17] -> (poses [3,5,N], bds [2,N]) as _load_data leaves them: the hwf column gets the size
    of the images actually used and the focal length divided by pose_factor (load_llff.py:62-64, 103-105)."""
    arr = np.asarray(poses_arr)
    poses = np.ascontiguousarray(np.moveaxis(arr[:, :15].reshape(-1, 3, 5), 0, -1))
    bds = np.ascontiguousarray(arr[:, 15:].T)
    poses[0, 4, :], poses[1, 4, :] = image_hw[0], image_hw[1]
    poses[2, 4, :] = poses[2, 4, :] * 1. / pose_factor
    return poses, bds


def llff_poses(poses, bds, recenter=True, bd_factor=.75, spherify=False, path_zflat=False):
    """The pose half of load_llff_data (load_llff.py:251-316).  poses [3,5,N], bds [2,N] as parse_poses_bounds returns
    them.  LLFF stores the camera axes as (down, right, back): they become (right, up, back); the scene is scaled so
    that the nearest depth bound is 1 / bd_factor; poses are recentred on their average; the render path is a two-turn
    spiral of 120 poses (half a flat turn of 60 with path_zflat) or, with spherify, the circle of spherify_poses.
    Returns (poses [N,3,5], bds [N,2], render_poses [M,3,5], i_test), arrays float32; i_test is the pose nearest to
    the average camera."""
    cams = np.moveaxis(np.asarray(poses), -1, 0).astype(np.float32)
    cams = np.concatenate([cams[:, :, 1:2], -cams[:, :, 0:1], cams[:, :, 2:]], 2)
    bds = np.ascontiguousarray(np.asarray(bds).T).astype(np.float32)
    if bd_factor is not None:
        scale = np.float32(1. / (bds.min() * bd_factor))
        cams[:, :3, 3] *= scale
        bds *= scale
    if recenter:
        cams = recenter_poses(cams)
    if spherify:
        cams, path, bds = spherify_poses(cams, bds)
    else:
        centre = poses_avg(cams)
        near, far = bds.min() * .9, bds.max() * 5.
        focus = 1. / (.25 / near + .75 / far)      # a depth between the bounds, weighted towards the near one in disparity
        radii = np.percentile(np.abs(cams[:, :3, 3]), 90, 0)
        turns, count = 2, 120
        if path_zflat:
            centre[:3, 3] += (-near * .1) * centre[:3, 2]
            radii[2] = 0.
            turns, count = 1, 60
        path = render_path_spiral(centre, _unit(cams[:, :3, 1].sum(0)), radii, focus, near * .2, zrate=.5, rots=turns,
                                  N=count)
    cams = np.asarray(cams, np.float32)
    nearest = int(np.argmin(np.sum((poses_avg(cams)[:3, 3] - cams[:, :3, 3]) ** 2, -1)))
    return cams, bds, np.asarray(path, np.float32), nearest


# ---------------------------------------------------------------------------- load_blender.py:15-34
def pose_spherical(theta, phi, radius):
    """4x4 camera-to-world on a sphere around the origin (degrees), as a float32 torch tensor like the reference's."""
    import torch
    from .synth import pose_spherical as _ps
    return torch.from_numpy(_ps(theta, phi, radius))


# ---------------------------------------------------------------------------- AVI (render_path's videos), no dependencies
def write_avi(path, frames8, fps=30, jpeg_quality=None):
    """frames8: uint8 [N,H,W,3] (or [N,H,W], written as grey RGB).  The reference writes its spiral / test-set videos
    with imageio.mimwrite(..., fps=30, quality=8) as MP4 (run_nerf.py:681, 819-820), which needs imageio + ffmpeg; this
    is the stand-in when they are absent: a RIFF/AVI file with one video stream, uncompressed 24-bit frames (any
    player, lossless), or Motion-JPEG frames when `jpeg_quality` (1..100) is given and Pillow is importable."""
    frames8 = np.ascontiguousarray(frames8)
    if frames8.dtype != np.uint8 or frames8.ndim not in (3, 4) or (frames8.ndim == 4 and frames8.shape[3] != 3):
        raise ValueError("write_avi expects uint8 [N,H,W,3] or [N,H,W], got %s %s" % (frames8.dtype, frames8.shape))
    if frames8.ndim == 3:
        frames8 = np.repeat(frames8[..., None], 3, axis=3)
    n, h, w = frames8.shape[:3]
    if n == 0:
        raise ValueError("write_avi: no frames")
    mjpg = False
    if jpeg_quality is not None:
        try:
            from PIL import Image
            import io
            mjpg = True
        except ImportError:
            mjpg = False
    chunks = []
    for fr in frames8:
        if mjpg:
            buf = io.BytesIO()
            Image.fromarray(fr).save(buf, format="JPEG", quality=int(jpeg_quality))
            data = buf.getvalue()
        else:      # bottom-up BGR rows, each padded to a multiple of 4 bytes
            rows = fr[::-1, :, ::-1].reshape(h, w * 3)
            pad = (-w * 3) % 4
            data = (np.concatenate([rows, np.zeros((h, pad), np.uint8)], 1) if pad else rows).tobytes()
        chunks.append(data)
    tag = b"00dc" if mjpg else b"00db"
    fourcc = b"MJPG" if mjpg else b"\0\0\0\0"
    biggest = max(len(c) for c in chunks)

    def riff_chunk(cid, body):
        return cid + struct.pack("<I", len(body)) + body + (b"\0" if len(body) & 1 else b"")

    def riff_list(kind, body):
        return b"LIST" + struct.pack("<I", len(body) + 4) + kind + body

    avih = struct.pack("<14I", int(round(1e6 / fps)), biggest * int(fps), 0, 0x10, n, 0, 1, biggest, w, h, 0, 0, 0, 0)
    strh = b"vids" + (fourcc if mjpg else b"DIB ") + struct.pack("<IHHIIIIIIIIhhhh", 0, 0, 0, 0, 1, int(fps), 0, n, biggest,
                                                                 0xffffffff, 0, 0, 0, w, h)
    strf = struct.pack("<IiiHH", 40, w, h, 1, 24) + fourcc + struct.pack("<IiiII", biggest if mjpg else len(chunks[0]), 0, 0, 0, 0)
    hdrl = riff_list(b"hdrl", riff_chunk(b"avih", avih) + riff_list(b"strl", riff_chunk(b"strh", strh) + riff_chunk(b"strf", strf)))
    movi_body, index, off = b"", b"", 4
    parts = []
    for c in chunks:
        parts.append(riff_chunk(tag, c))
        index += tag + struct.pack("<III", 0x10, off, len(c))
        off += len(parts[-1])
    movi_body = b"".join(parts)
    body = b"AVI " + hdrl + riff_list(b"movi", movi_body) + riff_chunk(b"idx1", index)
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", len(body)) + body)


def read_avi(path):
    """Reads back what write_avi wrote: (frames uint8 [N,H,W,3], fps).  Motion-JPEG frames need Pillow."""
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"RIFF" or data[8:12] != b"AVI ":
        raise ValueError("%s is not an AVI file" % path)
    w = h = fps = None
    frames = []

    def walk(lo, hi):
        nonlocal w, h, fps
        pos = lo
        while pos + 8 <= hi:
            cid, size = data[pos:pos + 4], struct.unpack("<I", data[pos + 4:pos + 8])[0]
            body = pos + 8
            if cid == b"LIST":
                walk(body + 4, body + size)
            elif cid == b"strh":
                fps = struct.unpack("<I", data[body + 24:body + 28])[0] / struct.unpack("<I", data[body + 20:body + 24])[0]
            elif cid == b"strf":
                w, h = struct.unpack("<ii", data[body + 4:body + 12])
            elif cid == b"00db":
                stride = (w * 3 + 3) // 4 * 4
                rows = np.frombuffer(data, np.uint8, stride * h, body).reshape(h, stride)[:, :w * 3]
                frames.append(rows.reshape(h, w, 3)[::-1, :, ::-1].copy())
            elif cid == b"00dc":
                from PIL import Image
                import io
                frames.append(np.asarray(Image.open(io.BytesIO(data[body:body + size])).convert("RGB")))
            pos = body + size + (size & 1)

    walk(12, len(data))
    return np.stack(frames, 0), fps


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
