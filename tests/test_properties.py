"""Property-based checks (hypothesis) of the host-side pieces every GPU result rests on: the Q6.12 token codec and the
`float2fix` writer (CNN.ipynb cell 23), the frame text format (cell 24), the batch partition of the multi-GPU drivers,
the sliding-window arithmetic of the SDR front-end, and the Q6.12 oracle's bit selection against Python integers."""
import numpy as np
from hypothesis import given, settings, strategies as st

from modulationdetectioncnn_amd.formats import q612
from modulationdetectioncnn_amd.frontend import window_count
from modulationdetectioncnn_amd.sharding import shard_bounds, shard_range
from oracle import oracle_q612 as OQ

Q = st.integers(min_value=-(1 << 17), max_value=(1 << 17) - 1)


@given(Q)
def test_q612_token_round_trip(v):
    bits = q612.int_to_bits(v)
    assert len(bits) == 18 and set(bits) <= {"0", "1"}
    assert q612.bits_to_int(bits, strict=True) == v
    # lenient decoding differs in ONE token: the hand-trimmed negative zero of the reference's writer bug (= -16.0 read
    # literally) means 0 (formats/q612.py header)
    assert q612.bits_to_int(bits) == (0 if bits == q612.NEGZERO_18 else v)
    # Verilog keeps the low 18 bits of an over-long sized literal
    assert q612.bits_to_int("0" + bits, strict=True) == v


@given(st.floats(min_value=-31.9990234375, max_value=31.9990234375, allow_nan=False, allow_infinity=False, width=32))
def test_float2fix_truncates_toward_zero_and_round_trips(x):
    """float2fix = sign-magnitude truncation of |x| * 2^12 (CNN.ipynb cell 23); the decoded value is within one LSB of x,
    never further from zero than x, and writing the decoded value again gives the same token."""
    tok = q612.float2fix(float(x))
    assert len(tok) == 18
    v = q612.bits_to_int(tok, strict=True)
    assert v == q612.float_to_int(float(x)) == int(float(x) * 4096)
    back = v / 4096.0
    assert abs(back) <= abs(x) and abs(x - back) < 1.0 / 4096.0
    assert q612.float2fix(back) == tok


@given(st.floats(min_value=-0.00024, max_value=-1e-12))
def test_float2fix_negative_zero_bug_and_its_repair(x):
    """small negative values truncate to magnitude 0: the reference's writer emits a 19-character token for them
    (bug_compatible=True reproduces it), this writer emits a plain zero; both decode to 0."""
    assert q612.float2fix(x) == "0" * 18
    bug = q612.float2fix(x, bug_compatible=True)
    assert len(bug) == 19 and q612.bits_to_int(bug) == 0


@settings(max_examples=25, deadline=None)
@given(st.lists(Q, min_size=256, max_size=256), st.booleans())
def test_frame_text_round_trip(ints, bug):
    """dump_frame -> parse: 256 rows `18'dNNN: data = 18'b...;`, I then Q (CNN.ipynb cell 24)"""
    frame = (np.array(ints, np.int32).reshape(2, 128).astype(np.float32) / np.float32(4096))
    text = q612.dump_frame(frame, bug_compatible=bug)
    p = q612.parse_text(text, strict=True)
    assert len(p.tables) == 1 and p.tables[0].indices == list(range(256))
    assert p.tables[0].rows == [int(v) for v in ints]
    lenient = q612.parse_text(text).tables[0].rows
    assert lenient == [0 if v == -65536 else int(v) for v in ints]


@given(st.integers(min_value=0, max_value=1 << 26), st.integers(min_value=1, max_value=64))
def test_shard_bounds_partition(n, world):
    b = shard_bounds(n, world)
    assert len(b) == world and b[0][0] == 0 and b[-1][1] == n
    assert all(lo <= hi for lo, hi in b) and all(b[i][1] == b[i + 1][0] for i in range(world - 1))
    sizes = [hi - lo for lo, hi in b]
    assert max(sizes) - min(sizes) <= 1
    assert [shard_range(n, r, world) for r in range(world)] == b


@given(st.integers(min_value=0, max_value=1 << 22), st.integers(min_value=1, max_value=4096))
def test_window_count_is_the_number_of_windows_that_fit(pairs, hop):
    if hop == 128:
        nbytes = (pairs // 128) * 256
        assert window_count(nbytes, hop) == pairs // 128
        return
    n = window_count(2 * pairs, hop)
    if n == 0:
        assert pairs < 128
    else:
        assert hop * (n - 1) + 128 <= pairs < hop * n + 128      # window n-1 fits, window n does not


@given(Q, Q, Q, Q)
def test_select18_is_the_verilog_bit_selection(a, b, c, d):
    """cnn_test_latest1.sv:653-655, 672-674 on Python integers: m = a*b + c*d as a 36-bit wire, out = {m[35], m[28:12]}"""
    m = (a * b + c * d) & ((1 << 36) - 1)
    want = (((m >> 35) & 1) << 17) | ((m >> 12) & 0x1FFFF)
    want = want - (1 << 18) if want & (1 << 17) else want
    got = int(OQ.select18(np.array([a * b + c * d], dtype=np.int64))[0])
    assert got == want
