"""Frame-parallel GOP sharding across the GPUs of one node.

Closed GOPs share nothing, and a JSV stream already carries a random-access index:
the START_MAP key map of 8-byte entries (absolute byte offset + timecode) that the
reference's seek relies on (decoders/jsv.js:264-268, :282-350, :1618-1648).  Rank 0
parses the stream header once and broadcasts the *stream index* -- sequence
parameters, quant matrices and the key map -- to every rank with ONE collective
(RCCL broadcast over xGMI; gloo in the CPU tests).  Rank r then decodes GOPs
{g : g mod world == r} with its own two-anchor reference state; no other data-path
collective exists (bitstream bytes reach a rank by file read, decoded frames stay on
the GPU that produced them).  An optional all-gather of per-GOP checksums serves
verification only.
"""
import numpy as np

MAGIC = 0x4C494458  # 'LIDX'
HEADER_WORDS = 10


def make_index(coded_w, coded_h, frame_w, frame_h, rate_idx, n_gops, gop_len, key_map=None,
               qm_intra=None, qm_non_intra=None):
    """Build the index rank 0 broadcasts.  key_map: [n_gops][2] uint32 (byte offset, timecode)."""
    if key_map is None:     # synthetic stream: evenly spaced entries
        km = np.zeros((n_gops, 2), dtype=np.uint32)
        km[:, 0] = np.arange(n_gops, dtype=np.uint64) * 65536 % (1 << 32)
        km[:, 1] = np.arange(n_gops, dtype=np.uint32) * gop_len
    else:
        km = np.ascontiguousarray(key_map, dtype=np.uint32).reshape(n_gops, 2)
    qi = np.zeros(64, np.uint8) if qm_intra is None else np.asarray(qm_intra, np.uint8)
    qn = np.zeros(64, np.uint8) if qm_non_intra is None else np.asarray(qm_non_intra, np.uint8)
    hdr = np.array([MAGIC, 1, coded_w, coded_h, frame_w, frame_h, rate_idx, n_gops, gop_len,
                    (qm_intra is not None) | ((qm_non_intra is not None) << 1)], dtype=np.uint32)
    blob = np.concatenate([hdr.view(np.uint8), qi, qn, km.view(np.uint8).reshape(-1)])
    return parse_index(blob)


def parse_index(blob):
    blob = np.ascontiguousarray(blob, dtype=np.uint8)
    hdr = blob[:4 * HEADER_WORDS].view(np.uint32)
    if hdr[0] != MAGIC or hdr[1] != 1:
        raise ValueError("not a leon stream index")
    n_gops = int(hdr[7])
    off = 4 * HEADER_WORDS
    qi = blob[off:off + 64].copy()
    qn = blob[off + 64:off + 128].copy()
    km = blob[off + 128:off + 128 + 8 * n_gops].view(np.uint32).reshape(n_gops, 2).copy()
    flags = int(hdr[9])
    return {"coded_w": int(hdr[2]), "coded_h": int(hdr[3]), "frame_w": int(hdr[4]), "frame_h": int(hdr[5]),
            "rate_idx": int(hdr[6]), "n_gops": n_gops, "gop_len": int(hdr[8]),
            "qm_intra": qi if flags & 1 else None, "qm_non_intra": qn if flags & 2 else None,
            "key_map": km, "blob": blob, "blob_bytes": int(blob.size)}


def broadcast_index(index, dist, torch, src=0):
    """One broadcast of the index blob from `src` (two messages: length, bytes).
    dist=None (single process) returns the index unchanged."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return index
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    rank = dist.get_rank()
    n = torch.tensor([index["blob_bytes"] if rank == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(n, src=src)
    if rank == src:
        buf = torch.from_numpy(index["blob"].copy()).to(dev)
    else:
        buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    dist.broadcast(buf, src=src)
    return parse_index(buf.cpu().numpy())


def shard_gops(index, rank, world):
    """GOP ids of this rank: round-robin over the key map, so every rank sees the same
    mix of stream positions and the per-GPU batch depth is identical (weak scaling)."""
    return [g for g in range(index["n_gops"]) if g % world == rank]


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return x ^ (x >> 31)


def gop_content(seed, gop_id, n_unique):
    """What GOP `gop_id` of a synthetic stream carries: a pure function of seed ^ gop_id, so that
    any rank -- or a single-rank run -- decoding the same GOP id decodes the same pictures.
    Returns (index of the synthetic GOP body among the n_unique generated ones, 56-bit key of the
    per-GOP variation applied on top, see gop_variation)."""
    h = _splitmix64((int(seed) ^ int(gop_id)) & 0xFFFFFFFFFFFFFFFF)
    return int(h % n_unique), int(h >> 8)


def gop_variation(key, n_mb, n=16):
    """The per-GOP variation: n macroblocks of the GOP's I picture get another quantiser scale
    (positions and values from the key).  Cheap to apply to a cloned map on the device, changes
    the decoded I picture and -- through prediction -- every picture of the GOP.
    Returns (macroblock indices int64[n], quantiser scales uint8[n])."""
    idx = np.array([_splitmix64(key + 2 * k + 1) % n_mb for k in range(n)], dtype=np.int64)
    val = np.array([1 + _splitmix64(key ^ (0xABCD + k)) % 31 for k in range(n)], dtype=np.uint8)
    return idx, val


def gather_checksums(local, dist, torch):
    """All-gather of this rank's per-GOP checksums (int64 vector) -> [world][n_local]."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return [list(map(int, local))]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor(list(map(int, local)), dtype=torch.int64, device=dev)
    out = [torch.zeros_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return [[int(v) for v in o.cpu().tolist()] for o in out]
