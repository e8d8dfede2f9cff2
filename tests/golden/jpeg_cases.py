"""Test inputs for the product's JPEG decoder: a small sequential-JPEG ENCODER (test code only)
that can produce the stream layouts the encoder vendored by the reference cannot - 4:2:2, 4:4:0,
4:1:1, restart intervals, one scan per component, 16-bit quantiser entries, grey scale - so that
tests/golden/make_golden.py can ask the reference's own loader what texels it makes of them.

The entropy code is deliberately not the standard one: every byte value gets a code (the first
128 values eight bits, the others nine), which is a valid canonical Huffman code and exercises
the decoder's table handling rather than a hard-wired table."""
import numpy as np
from scipy.fft import dctn

ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21,
          28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54,
          47, 55, 62, 63]


class _Bits:
    def __init__(self):
        self.out = bytearray()
        self.acc = 0
        self.n = 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            byte = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(byte)
            if byte == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):  # pad with ones to the byte boundary
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)


def _code(symbol):
    """(code, length) of the test's canonical code: bytes 0..127 -> 8 bits, 128..255 -> 9 bits."""
    return (symbol, 8) if symbol < 128 else (256 + (symbol - 128), 9)


def _dht(table_class, table_id):
    counts = [0] * 16
    counts[7], counts[8] = 128, 128
    body = bytes([(table_class << 4) | table_id]) + bytes(counts) + bytes(range(256))
    return b"\xFF\xC4" + (2 + len(body)).to_bytes(2, "big") + body


def _magnitude(v):
    size = int(abs(v)).bit_length()
    return size, (v if v >= 0 else v + (1 << size) - 1)


def _encode_block(bits, coef, pred):
    diff = int(coef[0]) - pred
    size, extra = _magnitude(diff)
    bits.put(*_code(size))
    if size:
        bits.put(extra, size)
    run = 0
    last = max([k for k in range(1, 64) if coef[k] != 0], default=0)
    for k in range(1, last + 1):
        if coef[k] == 0:
            run += 1
            continue
        while run > 15:
            bits.put(*_code(0xF0))
            run -= 16
        size, extra = _magnitude(int(coef[k]))
        bits.put(*_code((run << 4) | size))
        bits.put(extra, size)
        run = 0
    if last < 63:
        bits.put(*_code(0x00))
    return int(coef[0])


def encode(rgb, sampling=((2, 2), (1, 1), (1, 1)), quant_scale=4, restart_interval=0, interleaved=True,
           wide_quant=False, grey=False, rgb_ids=False):
    """A sequential Huffman JPEG of an (h, w, 3) uint8 image.  sampling: (H, V) per component."""
    h, w, _ = rgb.shape
    f = rgb.astype(np.float64)
    if rgb_ids:
        planes = [f[..., 0], f[..., 1], f[..., 2]]
    else:
        y = 0.299 * f[..., 0] + 0.587 * f[..., 1] + 0.114 * f[..., 2]
        cb = 128 - 0.168736 * f[..., 0] - 0.331264 * f[..., 1] + 0.5 * f[..., 2]
        cr = 128 + 0.5 * f[..., 0] - 0.418688 * f[..., 1] - 0.081312 * f[..., 2]
        planes = [y] if grey else [y, cb, cr]
    sampling = list(sampling)[:len(planes)]
    if grey:
        sampling = [(1, 1)]
    hmax, vmax = max(s[0] for s in sampling), max(s[1] for s in sampling)
    mcus_x, mcus_y = -(-w // (8 * hmax)), -(-h // (8 * vmax))
    quant = np.clip((np.add.outer(np.arange(8), np.arange(8)) + 2) * quant_scale, 1, 65535 if wide_quant else 255).astype(int)
    if wide_quant:
        quant[7, 7] = 300
    blocks = []  # per component: dict (by, bx) -> 64 quantised terms in zig-zag order
    sizes = []
    for p, (sh, sv) in zip(planes, sampling):
        fx, fy = hmax // sh, vmax // sv
        ph, pw = mcus_y * sv * 8, mcus_x * sh * 8
        # box-filter down, then pad by edge replication to whole MCUs
        full = np.pad(p, ((0, ph * fy - h), (0, pw * fx - w)), mode="edge")
        small = full.reshape(ph, fy, pw, fx).mean(axis=(1, 3))
        comp = {}
        for by in range(ph // 8):
            for bx in range(pw // 8):
                blk = small[8 * by:8 * by + 8, 8 * bx:8 * bx + 8] - 128.0
                c = np.rint(dctn(blk, norm="ortho") / quant).astype(int).reshape(64)
                comp[(by, bx)] = [c[ZIGZAG[k]] for k in range(64)]
        blocks.append(comp)
        sizes.append((-(-(-(-w * sh // hmax)) // 8), -(-(-(-h * sv // vmax)) // 8)))  # blocks that hold image samples

    out = bytearray(b"\xFF\xD8")
    if not rgb_ids:
        out += b"\xFF\xE0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00"
    qbody = bytearray([0x10 if wide_quant else 0x00])
    for k in range(64):
        v = int(quant.reshape(64)[ZIGZAG[k]])
        qbody += v.to_bytes(2, "big") if wide_quant else bytes([v])
    out += b"\xFF\xDB" + (2 + len(qbody)).to_bytes(2, "big") + qbody
    ids = [ord(c) for c in "RGB"] if rgb_ids else [1, 2, 3]
    sof = bytearray([8]) + h.to_bytes(2, "big") + w.to_bytes(2, "big") + bytes([len(planes)])
    for k, (sh, sv) in enumerate(sampling):
        sof += bytes([ids[k], (sh << 4) | sv, 0])
    out += b"\xFF\xC0" + (2 + len(sof)).to_bytes(2, "big") + sof
    out += _dht(0, 0) + _dht(1, 0)
    if restart_interval:
        out += b"\xFF\xDD\x00\x04" + restart_interval.to_bytes(2, "big")

    def scan(comps):
        nonlocal out
        hdr = bytearray([len(comps)])
        for k in comps:
            hdr += bytes([ids[k], 0x00])
        hdr += b"\x00\x3F\x00"
        out += b"\xFF\xDA" + (2 + len(hdr)).to_bytes(2, "big") + hdr
        bits = _Bits()
        pred = {k: 0 for k in comps}
        units = []
        if len(comps) == 1:
            k = comps[0]
            units = [[(k, by, bx)] for by in range(sizes[k][1]) for bx in range(sizes[k][0])]
        else:
            for my in range(mcus_y):
                for mx in range(mcus_x):
                    unit = []
                    for k in comps:
                        sh, sv = sampling[k]
                        unit += [(k, my * sv + yy, mx * sh + xx) for yy in range(sv) for xx in range(sh)]
                    units.append(unit)
        rst = 0
        for n, unit in enumerate(units):
            if restart_interval and n and n % restart_interval == 0:
                bits.flush()
                out += bits.out + bytes([0xFF, 0xD0 + rst])
                rst = (rst + 1) & 7
                bits = _Bits()
                pred = {k: 0 for k in comps}
            for k, by, bx in unit:
                pred[k] = _encode_block(bits, blocks[k][(by, bx)], pred[k])
        bits.flush()
        out += bits.out

    if interleaved or len(planes) == 1:
        scan(list(range(len(planes))))
    else:
        for k in range(len(planes)):
            scan([k])
    out += b"\xFF\xD9"
    return bytes(out)


def test_image(w, h, seed):
    """Smooth gradients plus some noise and an edge: every block has AC terms, chroma varies."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([x * 255.0 / max(w - 1, 1), y * 255.0 / max(h - 1, 1), 255.0 - (x + y) * 255.0 / max(w + h - 2, 1)], -1)
    img += rng.normal(0, 12, img.shape)
    img[h // 3:, w // 2:] = 255 - img[h // 3:, w // 2:]
    return np.clip(img, 0, 255).astype(np.uint8)


# name -> (w, h, encoder arguments)
CASES = {
    "s420_57x35": (57, 35, dict()),
    "s422_41x23": (41, 23, dict(sampling=((2, 1), (1, 1), (1, 1)))),
    "s440_23x41": (23, 41, dict(sampling=((1, 2), (1, 1), (1, 1)))),
    "s444_19x21": (19, 21, dict(sampling=((1, 1), (1, 1), (1, 1)))),
    "s411_70x18": (70, 18, dict(sampling=((4, 1), (1, 1), (1, 1)))),
    "s422_w2": (2, 9, dict(sampling=((2, 1), (1, 1), (1, 1)))),
    "s420_w1": (1, 1, dict()),
    "s420_restart3_64x48": (64, 48, dict(restart_interval=3)),
    "s422_restart1_33x17": (33, 17, dict(sampling=((2, 1), (1, 1), (1, 1)), restart_interval=1)),
    "s420_three_scans_40x40": (40, 40, dict(interleaved=False)),
    "s444_three_scans_restart_30x20": (30, 20, dict(sampling=((1, 1), (1, 1), (1, 1)), interleaved=False, restart_interval=2)),
    "s420_wide_quant_32x32": (32, 32, dict(wide_quant=True, quant_scale=2)),
    "grey_29x31": (29, 31, dict(grey=True)),
    "rgb_ids_24x24": (24, 24, dict(sampling=((1, 1), (1, 1), (1, 1)), rgb_ids=True)),
    "s420_coarse_48x48": (48, 48, dict(quant_scale=14)),
}
