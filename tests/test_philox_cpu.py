"""The numpy Philox4x32-10 restatement (tests/philox_ref.py: the checker of the device-side RANSAC index draw,
ransac-flow_amd/csrc/multih.hip) against the published known-answer vectors of the Random123 distribution, and the record
layout of the multi-homography drivers.  No GPU."""
import numpy as np

import philox_ref


def test_philox4x32_10_known_answers():
    kat = [([0, 0, 0, 0], (0, 0), [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, (0xffffffff, 0xffffffff), [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], (0xa4093822, 0x299f31d0), [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, out in kat:
        assert philox_ref.philox4x32_10(ctr, key).tolist() == out


def test_draw_mapping_is_modulo_the_count_and_keyed_by_pair_and_stream():
    s = philox_ref.draw_samples([1200, 5, 0], 64, seed=3, stream_id=(7 << 32) | 9)
    assert s.shape == (3, 64, 4) and s[0].max() < 1200 and s[1].max() < 5 and (s[2] == 0).all()
    raw = philox_ref.philox4x32_10([[10, 1, 9, 7]], (3, 0))[0].astype(np.int64)
    assert s[1, 10].tolist() == (raw % 5).tolist()
    assert not np.array_equal(s[0], philox_ref.draw_samples([1200], 64, seed=3, stream_id=8)[0])


def test_multih_record_layout():
    """One float32 row per pair: nbH | status | H (max_h,9) | flowDown8 | matchDown8 [| flowD2]; 16-byte aligned sections;
    0.85 MB / pair at 480x640 with 11 homographies."""
    from rfx import ops
    R = ops.MultiHRecords(3, 60, 80, "cpu", max_h=11)
    assert R.off_H == 4 and R.off_flow == 4 + 100 and R.off_match == R.off_flow + 11 * 2 * 4800
    assert R.width == R.off_match + 11 * 2 * 4800 and R.width % 4 == 0
    assert abs(R.width * 4 / 1e6 - 0.845) < 0.01
    nb, status, H, f8, m8, d2 = R.views()
    assert H.shape == (3, 11, 3, 3) and f8.shape == (3, 11, 2, 60, 80) and m8.shape == f8.shape and d2 is None
    assert float(status.sum()) == 3.0 and float(nb.sum()) == 0.0
    f8[1, 2, 1, 5, 7] = 3.0
    assert float(R.rec[1, R.off_flow + (2 * 2 + 1) * 4800 + 5 * 80 + 7]) == 3.0
    K = ops.MultiHRecords(2, 81, 268, "cpu", max_h=11, hd2=41, wd2=134)
    assert K.views()[5].shape == (2, 11, 2, 41, 134) and K.width == K.off_d2 + 11 * 2 * 41 * 134 + (-(K.off_d2 + 11 * 2 * 41 * 134)) % 4
