"""ORACLE — TEST INFRASTRUCTURE ONLY. Never imported by the product package.

CPU (numpy) restatement of the reference's detection decode:

  * get_topk_from_heatmap   <- centernet_lightning/models/centernet.py:243-261
  * gather_and_decode_boxes <- centernet_lightning/models/centernet.py:263-304
  * decode_detections       <- centernet_lightning/models/centernet.py:229-241
  * gather_at_indices       <- centernet_lightning/models/fairmot.py:63-73
  * gather_tracking2d       <- centernet_lightning/models/fairmot.py:138-151
  * pack / unpack of the all-gather record (the build's own wire format; the reference uses
    pickled objects, eval/coco.py:10-18)

Parity status: PINNED.  tests/golden/decode_*.npz hold outputs produced by the reference's own code
(imported in the build container by oracle/make_golden.py); tests/test_oracle_golden.py checks this
restatement against them bit-for-bit, and — when /root/reference is present — against the live
reference on fresh random inputs.

Tie rule.  torch.topk leaves the order of equal scores unspecified (SURVEY.md §7 "hard parts").
This oracle (and the HIP kernel) define the canonical order **(score descending, flat index
ascending)**; `canonicalize()` re-orders any reference output into that order so that tie groups
(e.g. the all-zero tail when an image has fewer than k peaks) compare equal.
"""
import numpy as np


def pseudo_nms(heat: np.ndarray, nms_kernel: int = 3) -> np.ndarray:
    """heat * (max_pool2d(heat, k, stride=1, pad=(k-1)//2) == heat); -inf padding, plateaus survive.
    centernet.py:249-253.  heat: (N, C, H, W) float32."""
    assert nms_kernel % 2 == 1, "reference breaks (shape mismatch) for even kernels"
    heat = np.asarray(heat, dtype=np.float32)
    p = (nms_kernel - 1) // 2
    H, W = heat.shape[-2:]
    padded = np.pad(heat, ((0, 0), (0, 0), (p, p), (p, p)), constant_values=-np.inf)
    m = np.full_like(heat, -np.inf)
    for dy in range(nms_kernel):
        for dx in range(nms_kernel):
            m = np.maximum(m, padded[..., dy:dy + H, dx:dx + W])
    mask = (m == heat)
    return heat * mask.astype(np.float32)


def get_topk_from_heatmap(heat: np.ndarray, num_detections: int = 100, nms_kernel: int = 3,
                          pseudo: bool = True):
    """centernet.py:243-261.  Returns scores (N,k) f32, indices (N,k) i64, labels (N,k) i64 in the
    canonical tie order."""
    heat = np.asarray(heat, dtype=np.float32)
    N = heat.shape[0]
    h = pseudo_nms(heat, nms_kernel) if pseudo else heat
    score = h.max(axis=1).reshape(N, -1)                 # torch.max(dim=1): values
    label = h.argmax(axis=1).reshape(N, -1)              # first maximal class (centernet.py:254)
    order = np.argsort(-score, axis=1, kind="stable")[:, :num_detections]   # (score desc, idx asc)
    scores = np.take_along_axis(score, order, axis=1)
    labels = np.take_along_axis(label, order, axis=1).astype(np.int64)
    return scores, order.astype(np.int64), labels


def gather_and_decode_boxes(box_offsets: np.ndarray, indices: np.ndarray, normalize_boxes: bool = False,
                            box_log: bool = False, box_multiplier: float = 1.0, stride: int = 4):
    """centernet.py:263-304.  box_offsets (N,4,H,W) ltrb in feature-map units -> (N,k,4) x1y1x2y2.
    All arithmetic in float32, one rounding per op (no fma contraction), like ATen."""
    box = np.asarray(box_offsets, dtype=np.float32)
    N, _, H, W = box.shape
    idx = np.asarray(indices, dtype=np.int64)
    cx = (idx % W).astype(np.float32) + np.float32(0.5)             # :278
    cy = (idx // W).astype(np.float32) + np.float32(0.5)            # :279
    flat = box.reshape(N, 4, H * W)
    g = np.stack([np.take_along_axis(flat[:, j], idx, axis=1) for j in range(4)], axis=-1)   # (N,k,4)
    if box_log:
        g = np.exp(g).astype(np.float32)                            # :283-284
    g = (g * np.float32(box_multiplier)).astype(np.float32)         # :285
    g = np.maximum(g, np.float32(0))                                # :286 clamp_min(0)
    x1 = cx - g[..., 0]
    y1 = cy - g[..., 1]
    x2 = cx + g[..., 2]
    y2 = cy + g[..., 3]
    boxes = np.stack([x1, y1, x2, y2], axis=-1).astype(np.float32)
    if normalize_boxes:
        boxes[..., [0, 2]] /= np.float32(W)                         # :299-301
        boxes[..., [1, 3]] /= np.float32(H)
    else:
        boxes *= np.float32(stride)                                 # :303
    return boxes


def gather_at_indices(reid: np.ndarray, indices: np.ndarray):
    """fairmot.py:63-73.  reid (N,E,H,W), indices (N,k) -> (N,k,E)."""
    reid = np.asarray(reid, dtype=np.float32)
    N, E = reid.shape[:2]
    flat = reid.reshape(N, E, -1)
    idx = np.broadcast_to(np.asarray(indices)[:, None, :], (N, E, indices.shape[1]))
    return np.take_along_axis(flat, idx, axis=2).swapaxes(1, 2).copy()


def decode_detections(heat, box_offsets, num_detections=100, nms_kernel=3, normalize_boxes=False,
                      box_log=False, box_multiplier=1.0, stride=4, reid=None):
    """centernet.py:229-241 (+ fairmot.py:138-151 when `reid` is given)."""
    scores, indices, labels = get_topk_from_heatmap(heat, num_detections, nms_kernel)
    boxes = gather_and_decode_boxes(box_offsets, indices, normalize_boxes, box_log, box_multiplier, stride)
    out = {"boxes": boxes, "scores": scores, "labels": labels, "indices": indices}
    if reid is not None:
        out["embeddings"] = gather_at_indices(reid, indices)
    return out


def decode_detections_torch(heat, box_offsets, num_detections=100, nms_kernel=3, normalize_boxes=False, box_multiplier=1.0, stride=4,
                            reid=None):
    """The same decode with the reference's own torch op sequence on CPU tensors (centernet.py:243-304: max_pool2d == heat mask,
    torch.max over classes, torch.topk, torch.gather; fairmot.py:63-73) — the multi-threaded form a CPU user of the reference runs.
    bench.py's cpu_baseline times THIS (the numpy functions above are single-threaded restatements kept for bit-exact checking);
    tests/test_oracle_golden.py checks both agree after canonicalize().  Torch's topk breaks score ties in an unspecified order."""
    import torch
    import torch.nn.functional as F
    N, _, H, W = heat.shape
    pad = (nms_kernel - 1) // 2
    m = F.max_pool2d(heat, nms_kernel, stride=1, padding=pad)
    h = heat * (m == heat)
    score, label = torch.max(h, dim=1)
    scores, indices = torch.topk(score.view(N, -1), num_detections)
    labels = torch.gather(label.view(N, -1), 1, indices)
    cx = (indices % W) + 0.5
    cy = torch.div(indices, W, rounding_mode="floor") + 0.5
    g = torch.gather(box_offsets.reshape(N, 4, -1), 2, indices.unsqueeze(1).expand(N, 4, num_detections)).swapaxes(1, 2)
    g = (g * box_multiplier).clamp_min(0)
    boxes = torch.stack([cx - g[..., 0], cy - g[..., 1], cx + g[..., 2], cy + g[..., 3]], dim=-1)
    if normalize_boxes:
        boxes[..., [0, 2]] /= W
        boxes[..., [1, 3]] /= H
    else:
        boxes = boxes * stride
    out = {"boxes": boxes, "scores": scores, "labels": labels, "indices": indices}
    if reid is not None:
        E = reid.shape[1]
        out["embeddings"] = torch.gather(reid.reshape(N, E, -1), 2, indices.unsqueeze(1).expand(N, E, num_detections)).swapaxes(1, 2)
    return out


def canonicalize(scores, indices, *others):
    """Re-order each image's detections into (score desc, index asc).  Only permutes inside groups of
    equal score, so it is the identity on tie-free outputs."""
    scores = np.asarray(scores)
    indices = np.asarray(indices)
    order = np.lexsort((indices, -scores.astype(np.float64)), axis=1)
    take = lambda a: np.take_along_axis(a, order.reshape(order.shape + (1,) * (a.ndim - 2)), axis=1)
    return (take(scores), take(indices)) + tuple(take(np.asarray(o)) for o in others)


# ----------------------------------------------------------------------------------------------
# all-gather record: one float32 row per detection  [x1, y1, x2, y2, score, label(bits), emb...]
# ----------------------------------------------------------------------------------------------
def pack_detections(boxes, scores, labels, embeddings=None):
    """(N,k,4) f32, (N,k) f32, (N,k) i64 [, (N,k,E) f32] -> (N,k,6[+E]) f32.  The label travels as the
    bit pattern of its low 32 bits (int32), not as a converted float, so it round-trips exactly."""
    N, k = scores.shape
    E = 0 if embeddings is None else embeddings.shape[-1]
    rec = np.empty((N, k, 6 + E), dtype=np.float32)
    rec[..., 0:4] = boxes
    rec[..., 4] = scores
    rec[..., 5] = labels.astype(np.int32).view(np.float32)
    if E:
        rec[..., 6:] = embeddings
    return rec


def unpack_detections(rec):
    rec = np.ascontiguousarray(rec, dtype=np.float32)
    out = {"boxes": rec[..., 0:4].copy(), "scores": rec[..., 4].copy(),
           "labels": rec[..., 5].copy().view(np.int32).astype(np.int64)}
    if rec.shape[-1] > 6:
        out["embeddings"] = rec[..., 6:].copy()
    return out


# ----------------------------------------------------------------------------------------------
# pre-processing before the path
# ----------------------------------------------------------------------------------------------
def normalize_u8(images_u8, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225), max_pixel_value=255.0):
    """albumentations A.Normalize restated (third-party, absent from the reference tree; called at README.md:83-86,
    datasets/utils.py:13-15 with the IMAGENET constants of datasets/utils.py:9-10):
        mean *= max_pixel; std *= max_pixel; denominator = reciprocal(std); img = float32(img); img -= mean; img *= denominator
    all float32.  images_u8: (N,H,W,3) uint8 -> (N,H,W,3) float32.  "parity unpinned" by the reference's tests (it has none for
    pre-processing); the published algorithm is restated."""
    m = np.array(mean, dtype=np.float32) * np.float32(max_pixel_value)
    s = np.array(std, dtype=np.float32) * np.float32(max_pixel_value)
    denom = np.reciprocal(s, dtype=np.float32)
    img = np.asarray(images_u8).astype(np.float32)
    img = img - m
    img = img * denom
    return img.astype(np.float32)


def resize_bilinear_u8(images_u8, out_h, out_w):
    """cv2.resize(img, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for uint8 images = albumentations A.Resize (README.md:84 of the
    reference).  OpenCV is absent here: its published 8-bit algorithm (modules/imgproc/src/resize.cpp: resizeGeneric_ with
    HResizeLinear<uchar,int,short,2048> and VResizeLinear<uchar,int,short>) is restated — "parity unpinned":
        fx = float32((dx + 0.5) * scale_x - 0.5); sx = floor(fx); fx -= sx; clamp (sx < 0 -> 0, fx = 0; sx >= W-1 -> W-1, fx = 0)
        alpha = int16(rint([1 - fx, fx] * 2048)); beta likewise from fy (rows clipped instead of fy clamped)
        D = S[sx] * a0 + S[min(sx+1, W-1)] * a1;   dst = (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2
    images_u8: (N,H,W,C) uint8 -> (N,out_h,out_w,C) uint8."""
    x = np.asarray(images_u8)
    N, H, W, C = x.shape
    sx_scale, sy_scale = 1.0 / (out_w / W), 1.0 / (out_h / H)

    def axis(n_out, n_in, scale, clamp_f):
        d = np.arange(n_out, dtype=np.float64)
        f = ((d + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = (f - s.astype(np.float32)).astype(np.float32)
        if clamp_f:
            lo, hi = s < 0, s >= n_in - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, n_in - 1, s))
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int16).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int16).astype(np.int64)
        return s, c0, c1

    sx, a0, a1 = axis(out_w, W, sx_scale, True)
    sy, b0, b1 = axis(out_h, H, sy_scale, False)
    x1 = np.minimum(sx + 1, W - 1)
    y0, y1 = np.clip(sy, 0, H - 1), np.clip(sy + 1, 0, H - 1)
    xi = x.astype(np.int64)
    D = xi[:, :, sx, :] * a0[None, None, :, None] + xi[:, :, x1, :] * a1[None, None, :, None]          # (N, H, out_w, C)
    D0, D1 = D[:, y0], D[:, y1]
    v = (((b0[None, :, None, None] * (D0 >> 4)) >> 16) + ((b1[None, :, None, None] * (D1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)
