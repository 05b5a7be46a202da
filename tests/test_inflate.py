"""The reader's own DEFLATE decoder (c3poa_amd/csrc/c3_inflate.hpp) against zlib: every block type, every level / strategy / window size,
history across chunk borders, damaged streams.  CPU only (the decoder is host code; the library loads without a GPU)."""
import ctypes as C
import gzip
import os
import random
import zlib

import numpy as np
import pytest

from c3poa_amd import _lib


def _read_all(path):
    rd = _lib.Reader(path, n_sets=2)
    out = []
    while True:
        hb = rd.next(700)
        if hb.n == 0:
            break
        out += [hb.read(i) for i in range(hb.n)]
    rd.close()
    return out


def _inflate(raw, size, chunk):
    lib = _lib.load()
    lib.c3_debug_inflate.restype = C.c_long
    lib.c3_debug_inflate.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_size_t]
    out = C.create_string_buffer(size + 1)
    n = lib.c3_debug_inflate(raw, len(raw), out, size, chunk)
    return n, out.raw[:max(n, 0)]


def _deflate(data, level, strategy=zlib.Z_DEFAULT_STRATEGY, wbits=15, flush_every=0):
    co = zlib.compressobj(level, zlib.DEFLATED, -wbits, 9, strategy)
    if not flush_every:
        return co.compress(data) + co.flush()
    out = b""
    for i in range(0, len(data), flush_every):          # sync / full flushes: empty stored blocks, byte alignment in mid-stream
        out += co.compress(data[i:i + flush_every]) + co.flush(zlib.Z_FULL_FLUSH if (i // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
    return out + co.flush()


def _payloads(seed):
    rng = random.Random(seed)
    nprng = np.random.default_rng(seed)
    fq = []
    for i in range(400):
        n = rng.randint(50, 3000)
        seq = "".join(rng.choice("ACGT") for _ in range(n))
        if i % 7 == 0:
            seq = seq[:n // 3] + "A" * (n // 3) + "AT" * (n // 6)                       # homopolymers, short periods (distance 1 / 2 matches)
        q = "".join(chr(rng.randint(33, 73)) for _ in range(len(seq))) if i % 3 else "I" * len(seq)
        fq.append("@read%d\n%s\n+\n%s\n" % (i, seq, q))
    yield ("".join(fq)).encode()
    yield bytes(nprng.integers(0, 256, 300000, dtype=np.uint8))                          # incompressible: stored blocks / literals only
    yield bytes(nprng.integers(0, 4, 200000, dtype=np.uint8))                            # tiny alphabet: long codes on rare symbols
    yield b"\x00" * 100000 + b"abc" * 30000 + bytes(range(256)) * 50                     # runs, period 3, all byte values
    yield b""                                                                            # an empty stream
    yield b"x"
    # far matches: a block repeated at the maximum distance
    blk = bytes(nprng.integers(0, 256, 32768, dtype=np.uint8))
    yield blk + blk + blk[:5000]
    # concatemers (what the CLI reads): every read repeats a unit with errors, so copies refer to copies of copies -- with `gzip -1` a quarter to
    # half of a chunk decoded without its window stays "a byte of the unknown window" to its end (the parallel decoder's index plane)
    cc = []
    for i in range(120):
        unit = "".join(rng.choice("ACGT") for _ in range(rng.randint(60, 500)))
        n = rng.randint(500, 5000)
        seq = "".join(unit[k % len(unit)] if rng.random() > 0.1 else rng.choice("ACGT") for k in range(n))
        cc.append("@c%07d ch=%d\n%s\n+\n%s\n" % (i, rng.randint(1, 512), seq, "".join(chr(rng.randint(34, 60)) for _ in range(n))))
    yield ("".join(cc)).encode()


@pytest.mark.parametrize("seed", [1, 2])
def test_every_kind_of_stream_matches_zlib(seed):
    n_cases = 0
    for data in _payloads(seed):
        for level, strategy, wbits, fe in [(0, zlib.Z_DEFAULT_STRATEGY, 15, 0), (1, zlib.Z_DEFAULT_STRATEGY, 15, 0), (6, zlib.Z_DEFAULT_STRATEGY, 15, 0),
                                           (9, zlib.Z_DEFAULT_STRATEGY, 15, 0), (6, zlib.Z_FIXED, 15, 0), (6, zlib.Z_HUFFMAN_ONLY, 15, 0), (6, zlib.Z_RLE, 15, 0),
                                           (6, zlib.Z_FILTERED, 12, 0), (9, zlib.Z_DEFAULT_STRATEGY, 9, 0), (5, zlib.Z_DEFAULT_STRATEGY, 15, 7001)]:
            raw = _deflate(data, level, strategy, wbits, fe)
            for chunk in (0, 1 << 16, 40000, 1 << 20):
                n, got = _inflate(raw, len(data), chunk)
                assert n == len(data) and got == data, (len(data), level, strategy, wbits, fe, chunk, n)
                n_cases += 1
    assert n_cases > 200


def test_damaged_streams_are_errors_or_caught_by_the_length():
    rng = random.Random(5)
    data = ("".join("@r%d\n%s\n+\n%s\n" % (i, "".join(rng.choice("ACGT") for _ in range(500)), "5" * 500) for i in range(200))).encode()
    raw = _deflate(data, 6)
    bad = 0
    for k in range(300):
        b = bytearray(raw)
        pos = rng.randrange(len(b)); b[pos] ^= 1 << rng.randrange(8)
        n, got = _inflate(bytes(b), len(data) + 70000, 1 << 16)
        # a flipped bit either breaks the stream (an error), changes its length, or changes bytes (the reader's CRC-32 check catches those)
        if n < 0 or got != data:
            bad += 1
    assert bad >= 295
    # truncation is always an error
    for cut in (1, 2, 10, len(raw) // 2, len(raw) - 1):
        n, _ = _inflate(raw[:cut], len(data), 0)
        assert n < 0


def test_gz_files_of_every_layout_through_the_reader(tmp_path):
    """plain gzip files (one member, several members, header fields, trailing zeros) read by c3_reader with the own decoder"""
    rng = random.Random(11)
    recs = [("r%d" % i, "".join(rng.choice("ACGT") for _ in range(rng.randint(100, 4000))), None) for i in range(1500)]
    recs = [(n, s, "".join(chr(rng.randint(40, 70)) for _ in s)) for n, s, _ in recs]
    text = "".join("@%s\n%s\n+\n%s\n" % r for r in recs).encode()
    third = len(text) // 3
    cut1 = text.rfind(b"\n@", 0, third) + 1; cut2 = text.rfind(b"\n@", 0, 2 * third) + 1
    layouts = {
        "one.gz": gzip.compress(text, 6),
        "three.gz": gzip.compress(text[:cut1], 1) + gzip.compress(text[cut1:cut2], 9) + gzip.compress(text[cut2:], 0),
        "pad.gz": gzip.compress(text, 4) + b"\0" * 512,
    }
    import io
    buf = io.BytesIO()
    with gzip.GzipFile(filename="reads_with_a_name.fastq", mode="wb", fileobj=buf, compresslevel=7) as g:
        g.write(text)
    layouts["named.gz"] = buf.getvalue()
    for name, blob in layouts.items():
        p = tmp_path / name
        p.write_bytes(blob)
        assert _read_all(str(p)) == recs, name
    # a member whose CRC is wrong
    blob = bytearray(layouts["one.gz"]); blob[-6] ^= 0x40
    p = tmp_path / "crc.gz"; p.write_bytes(bytes(blob))
    with pytest.raises(ValueError):
        _read_all(str(p))


def test_a_text_file_named_gz_and_an_empty_gz(tmp_path):
    """gzread passes a file through that is not gzip at all: that behaviour stays; an empty gzip member is an empty input"""
    recs = [("a", "ACGTACGT", "IIIIIIII"), ("b", "TTTT", "5555")]
    p = tmp_path / "plain.fastq.gz"
    p.write_text("".join("@%s\n%s\n+\n%s\n" % r for r in recs))
    assert _read_all(str(p)) == recs
    p2 = tmp_path / "empty.fastq.gz"
    p2.write_bytes(gzip.compress(b""))
    assert _read_all(str(p2)) == []


def test_the_deepest_huffman_trees():
    """Fibonacci-like symbol frequencies force 12- to 15-bit codes: the second-level tables"""
    rng = random.Random(3)
    fib = [1, 1]
    while len(fib) < 32:
        fib.append(fib[-1] + fib[-2])
    for nsym in (20, 26, 31):
        syms = []
        for i in range(nsym):
            syms += [i * 5 + 3] * min(fib[i], 60000)
        rng.shuffle(syms)
        data = bytes(syms)
        for strategy in (zlib.Z_HUFFMAN_ONLY, zlib.Z_DEFAULT_STRATEGY):
            raw = _deflate(data, 9, strategy)
            for chunk in (0, 50000):
                n, got = _inflate(raw, len(data), chunk)
                assert n == len(data) and got == data, (nsym, strategy, chunk, n)


# ---- round 6: ONE gzip stream inflated by several threads (c3poa_amd/csrc/c3_gzpar.hpp) -----------------------------------------------
def _gunzip_par(gz, threads, chunk, cap):
    lib = _lib.load()
    lib.c3_debug_gunzip_par.restype = C.c_long
    lib.c3_debug_gunzip_par.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_size_t, C.c_char_p, C.c_size_t]
    out = C.create_string_buffer(cap + 1)
    n = lib.c3_debug_gunzip_par(gz, len(gz), threads, chunk, out, cap)
    return n, out.raw[:max(n, 0)]


def _gz(data, level=6, flush_every=0):
    if not flush_every:
        return gzip.compress(data, compresslevel=level)
    co = zlib.compressobj(level, zlib.DEFLATED, 31)
    out = b""
    for i in range(0, len(data), flush_every):          # pigz-like: sync / full flushes (empty stored blocks, the window dropped)
        out += co.compress(data[i:i + flush_every]) + co.flush(zlib.Z_FULL_FLUSH if (i // flush_every) % 2 else zlib.Z_SYNC_FLUSH)
    return out + co.flush()


@pytest.mark.parametrize("seed", [1, 2])
def test_one_gzip_stream_by_several_threads_matches_zlib(seed):
    """every payload x level x (threads, chunk): chunks from a few KB (every chunk seam inside a few blocks: markers across many seams,
    chunks without any block start, block starts found in the middle of members) to larger than the file (one chunk)"""
    for data in _payloads(seed):
        for level in (1, 6, 9, 0):
            gz = _gz(data, level)
            for threads, chunk in ((4, 3000), (3, 20000), (8, 70000), (2, 1 << 22)):
                n, out = _gunzip_par(gz, threads, chunk, len(data))
                assert n == len(data) and out == data, (seed, level, threads, chunk, n, len(data))
        gz = _gz(data, 6, flush_every=50000)
        n, out = _gunzip_par(gz, 4, 9000, len(data))
        assert n == len(data) and out == data


def test_parallel_gunzip_members_trailing_bytes_and_damage():
    data = next(iter(_payloads(5)))
    third = len(data) // 3
    mm = _gz(data[:third], 6) + _gz(b"", 6) + _gz(data[third:2 * third], 1) + _gz(data[2 * third:], 9)      # concatenated members (one stream, as for gzread), an empty one among them
    for threads, chunk in ((4, 5000), (8, 40000), (2, 1 << 22)):
        n, out = _gunzip_par(mm, threads, chunk, len(data))
        assert n == len(data) and out == data
    n, out = _gunzip_par(mm + b"not a gzip member", 4, 5000, len(data))                                       # trailing bytes end the input (zlib: ignored)
    assert n == len(data) and out == data
    assert _gunzip_par(b"plain text, no gzip", 4, 5000, 100)[0] == -3
    gz = _gz(data, 6)
    rng = random.Random(9)
    caught = 0
    for _ in range(60):                                                                                       # one flipped bit anywhere: an error, or (a flip inside the header's name / time fields) the same bytes
        d = bytearray(gz)
        k = rng.randrange(len(d))
        d[k] ^= 1 << rng.randrange(8)
        n, out = _gunzip_par(bytes(d), 4, 6000, len(data) + 70000)
        assert n < 0 or out == data, k
        caught += n < 0
    assert caught >= 50
    for cut in (10, 1000, len(gz) // 2):                                                                      # a truncated file is an error, never a shorter stream
        assert _gunzip_par(gz[:-cut], 4, 6000, len(data))[0] == -1


def test_plain_gz_file_through_the_reader_with_several_inflating_threads(tmp_path, monkeypatch):
    """the reader picks the parallel decoder for a plain .gz of at least two chunks (C3_GZ_CHUNK shrinks them for the test); the records
    equal those of the uncompressed file, for one thread (the serial decoder), three and eight; a damaged file is an error"""
    rng = random.Random(4)
    recs = []
    for i in range(1500):
        n = rng.randint(200, 6000)
        recs.append(("r%d some description" % i, "".join(rng.choice("ACGT") for _ in range(n)), "".join(chr(rng.randint(35, 70)) for _ in range(n))))
    text = "".join("@%s\n%s\n+\n%s\n" % r for r in recs).encode()
    plain = tmp_path / "a.fastq"
    plain.write_bytes(text)
    want = _read_all(str(plain))
    gzp = tmp_path / "a.fastq.gz"
    gzp.write_bytes(gzip.compress(text, 6))
    monkeypatch.setenv("C3_GZ_CHUNK", "40000")
    for th in ("1", "3", "8"):
        monkeypatch.setenv("C3_GZ_THREADS", th)
        assert _read_all(str(gzp)) == want
    d = bytearray(gzp.read_bytes())
    d[len(d) // 2] ^= 0x10
    bad = tmp_path / "bad.fastq.gz"
    bad.write_bytes(bytes(d))
    monkeypatch.setenv("C3_GZ_THREADS", "4")
    with pytest.raises(Exception):
        _read_all(str(bad))


def test_plain_gz_with_default_chunks_hands_buffers_over_without_copying(tmp_path, monkeypatch):
    """a file of several default-size (1 MiB) chunks: the parser takes every chunk's buffer as it is (gzpar_swap: the partial line at the end
    of one buffer moves into the free bytes in front of the next) -- records equal the plain file's, lines of 30 kb among them (a rest
    longer than usual) and one line longer than the free space in front of a chunk (that refill copies instead)"""
    rng = random.Random(12)
    recs = []
    for i in range(1400):
        n = rng.choice([3000, 9000, 30000]) + rng.randint(0, 500)
        if i == 700:
            n = 700000                                     # longer than the 512 KiB in front of a chunk
        recs.append(("q%d" % i, "".join(rng.choice("ACGT") for _ in range(n)), "".join(chr(rng.randint(40, 60)) for _ in range(n))))
    text = "".join("@%s\n%s\n+\n%s\n" % r for r in recs).encode()
    plain = tmp_path / "b.fastq"
    plain.write_bytes(text)
    gzp = tmp_path / "b.fastq.gz"
    gzp.write_bytes(gzip.compress(text, 1))
    assert gzp.stat().st_size > 4 << 20
    want = _read_all(str(plain))
    for th in ("4", "2"):
        monkeypatch.setenv("C3_GZ_THREADS", th)
        assert _read_all(str(gzp)) == want


def test_the_readers_crc32_is_zlibs():
    """c3_crc32.hpp (carry-less multiplication, constants derived from the polynomial) against zlib.crc32: every length 0..400 at every
    alignment 0..16, continued from arbitrary states, and long buffers."""
    lib = _lib.load()
    lib.c3_debug_crc32.restype = C.c_uint
    lib.c3_debug_crc32.argtypes = [C.c_uint, C.c_char_p, C.c_size_t]
    rng = np.random.default_rng(77)
    buf = rng.integers(0, 256, size=3 << 20, dtype=np.uint8).tobytes()
    for n in range(0, 401):
        for a in range(0, 17):
            s = int(rng.integers(0, 1 << 32))
            piece = buf[a:a + n]
            assert lib.c3_debug_crc32(s, piece, n) == zlib.crc32(piece, s), (n, a)
    for _ in range(60):
        n, a, s = int(rng.integers(0, 3 << 20)), int(rng.integers(0, 64)), int(rng.integers(0, 1 << 32))
        piece = buf[a:a + n]
        assert lib.c3_debug_crc32(s, piece, len(piece)) == zlib.crc32(piece, s), (n, a)
    # in pieces = in one go
    c = 0
    for k in range(0, len(buf), 100003):
        c = lib.c3_debug_crc32(c, buf[k:k + 100003], len(buf[k:k + 100003]))
    assert c == zlib.crc32(buf)
