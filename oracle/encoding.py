"""Oracle (CPU checker, test infrastructure only): spatio-temporal encoding of
/root/reference/busca/encodings.py restated with torch CPU ops in the reference's own dtype flow.

The reference materialises a 211x211x61xd fp16 table (encodings.py:23-32).  The table is separable
per axis, so the oracle keeps three LUTs and concatenates rows; the golden tests
check the LUT rows against rows of the real table built by the reference.
"""
import numpy as np
import torch

MAX_TEMP = 30       # encodings.py:10
MAX_DIST = 105
MAX_SIZE = 105
F32_MIN = float(np.finfo(np.float32).min)


def axis_channels(d):
    """positional_encodings 6.0.3 PositionalEncoding3D.__init__: c = 2*ceil(d/6) (made even)."""
    c = int(np.ceil(d / 6) * 2)
    if c % 2:
        c += 1
    return c


def build_luts(d):
    """Three fp16 LUTs [211,c], [211,c], [61,c]: interleaved sin/cos of pos * inv_freq computed in
    float32 with torch CPU ops (the same ops the third-party module uses), then rounded to fp16
    (encodings.py:28-31).  Channel layout of an encoding row: [xy LUT | size LUT | time LUT][:d]."""
    c = axis_channels(d)
    inv_freq = 1.0 / (10000 ** (torch.arange(0, c, 2).float() / c))

    def lut(n):
        pos = torch.arange(n, dtype=inv_freq.dtype)
        s = torch.einsum("i,j->ij", pos, inv_freq)
        e = torch.stack((s.sin(), s.cos()), dim=-1).flatten(-2, -1)
        return e.to(torch.float16)

    return lut(2 * MAX_DIST + 1), lut(2 * MAX_SIZE + 1), lut(2 * MAX_TEMP + 1)


def encoding_rows(luts, ixy, isz, it, d):
    """pe[ixy, isz, it] (encodings.py:71,81) from the separable LUTs -> float32 [..., d]."""
    lx, ls, lt = luts
    row = torch.cat([lx[ixy.long()], ls[isz.long()], lt[it.long()]], dim=-1)[..., :d]
    return row.float()


def distant_fake_bbox(f64):
    """encodings.py:21: torch.from_numpy(missing_candidate_bbox('ltwh')) - float64 under the reference's
    pinned numpy 1.23.5, float32 under numpy >= 2."""
    if f64:
        v = [F32_MIN, F32_MIN, -F32_MIN / 100.0, -F32_MIN / 100.0]
        return torch.tensor(v, dtype=torch.float64)
    m = np.float32(F32_MIN)
    return torch.from_numpy(np.array([m, m, -m / np.float32(100.0), -m / np.float32(100.0)], dtype=np.float32))


FLAVOURS = ("MEM-SEP-CAN-BAD", "MEM-SEP-CAN", "MEM-CAN-SEP-BAD", "MEM-CAN-SEP")   # the CLS-* ones fail inside the reference itself (encodings.py:161)


def insert_fake_bboxes(can_bboxes, ref_bbox, fake_f64=True, encode_sep_as_ref=True, flavour="MEM-SEP-CAN-BAD"):
    """encodings.py:97-148.  MEM-SEP-CAN*: [ref|can_i]*P, [ref, ref] (SEP, NON) and, with -BAD, [fake, fake] (SEP, BAD);
    MEM-CAN-SEP*: [can_i|ref]*P, ...; encode_sep_as_ref=False encodes a separator with its candidate's box.  torch.cat promotes
    to float64 when the (float64) fake bbox takes part - i.e. only in the -BAD flavours."""
    B, P, _ = can_bboxes.shape
    parts = []
    for i in range(P):
        sep = ref_bbox if encode_sep_as_ref else can_bboxes[:, [i], :]
        parts += [sep, can_bboxes[:, [i], :]] if "MEM-SEP-CAN" in flavour else [can_bboxes[:, [i], :], sep]
    parts += [ref_bbox, ref_bbox]
    if "BAD" in flavour:
        fake = distant_fake_bbox(fake_f64).repeat(B, 1, 1)
        parts += [fake, fake]
    return torch.cat(parts, dim=1)


def extract_distance_values(bbox, ref_bbox):
    """encodings.py:238-272, op for op."""
    xmin, ymin, xmax, ymax = torch.tensor_split(ref_bbox, 4, dim=1)
    w_ref = xmax - xmin + 1
    h_ref = ymax - ymin + 1
    cx_ref = 0.5 * (xmin + xmax)
    cy_ref = 0.5 * (ymin + ymax)
    xmin, ymin, xmax, ymax = torch.tensor_split(bbox, 4, dim=1)
    w = xmax - xmin + 1
    h = ymax - ymin + 1
    cx = 0.5 * (xmin + xmax)
    cy = 0.5 * (ymin + ymax)
    dx = torch.pow((cx - cx_ref) / w, 2)
    dy = torch.pow((cy - cy_ref) / h, 2)
    xy = (torch.sqrt(dx + dy) + 1e-3).log()
    dw = (w / w_ref + 1e-3).log()
    dh = (h / h_ref + 1e-3).log()
    return xy, dw + dh


def temporal_ids(L, P2, range_factor=2.0):
    """encodings.py:150-180: mem -> clamp((i-L+1)*2, +-30)+30 ; candidate tokens [1,2]*(P+2) -> *2 + 30."""
    mem = torch.tensor(list(range(-L + 1, 1)))
    can = torch.tensor([1, 2] * P2)
    mem = torch.clamp(mem * range_factor, min=-MAX_TEMP, max=MAX_TEMP).to(torch.long) + MAX_TEMP
    can = torch.clamp(can * range_factor, min=-MAX_TEMP, max=MAX_TEMP).to(torch.long) + MAX_TEMP
    return mem, can


def spatial_ids(mem_bboxes, can_bboxes_with_fakes, range_factor=15.0):
    """encodings.py:183-235: clamp(v*15, +-105).to(long) + 105 (truncation toward zero)."""
    B = mem_bboxes.shape[0]
    ref = mem_bboxes[:, -1:, :]
    ref_can = ref.repeat(1, can_bboxes_with_fakes.shape[1], 1).view(-1, 4)
    ref_mem = ref.repeat(1, mem_bboxes.shape[1], 1).view(-1, 4)
    cxy, csz = extract_distance_values(can_bboxes_with_fakes.reshape(-1, 4), ref_can)
    cxy = torch.clamp(cxy.view(B, -1) * range_factor, min=-MAX_DIST, max=MAX_DIST).to(torch.long)
    csz = torch.clamp(csz.view(B, -1) * range_factor, min=-MAX_SIZE, max=MAX_SIZE).to(torch.long)
    csz = torch.clamp(csz, min=-MAX_SIZE, max=MAX_SIZE).to(torch.long)
    mxy, msz = extract_distance_values(mem_bboxes.reshape(-1, 4), ref_mem)
    mxy = torch.clamp(mxy.view(B, -1) * range_factor, min=-MAX_DIST, max=MAX_DIST).to(torch.long)
    msz = torch.clamp(msz.view(B, -1) * range_factor, min=-MAX_SIZE, max=MAX_SIZE).to(torch.long)
    return (mxy + MAX_DIST, msz + MAX_SIZE), (cxy + MAX_DIST, csz + MAX_SIZE)


def token_bucket_ids(mem_bboxes, can_bboxes, fake_f64=True, flavour="MEM-SEP-CAN-BAD", encode_sep_as_ref=True):
    """All three bucket indices per token, token order [MEM*L, pair*P, pair(NON) [, pair(BAD)]] with pair = (SEP, CAN) or (CAN, SEP).
    Returns int64 [B, T, 3] with columns (xy, size, time)."""
    mem_bboxes = torch.as_tensor(mem_bboxes, dtype=torch.float32)
    can_bboxes = torch.as_tensor(can_bboxes, dtype=torch.float32)
    B, L, _ = mem_bboxes.shape
    P = can_bboxes.shape[1]
    ref = mem_bboxes[:, -1:, :].clone()
    fakes = insert_fake_bboxes(can_bboxes, ref, fake_f64=fake_f64, encode_sep_as_ref=encode_sep_as_ref, flavour=flavour)
    (mxy, msz), (cxy, csz) = spatial_ids(mem_bboxes, fakes)
    mt, ct = temporal_ids(L, P + (2 if "BAD" in flavour else 1))
    xy = torch.cat([mxy, cxy], dim=1)
    sz = torch.cat([msz, csz], dim=1)
    t = torch.cat([mt, ct]).repeat(B, 1)
    return torch.stack([xy, sz, t], dim=-1)
