"""Host-side bookkeeping of the enqueue-ahead path (no GPU): the depth-key bit hint follows a slowly decaying
maximum (cameras whose depth ranges differ by a bit or two must not alternate between a miss and a reset)."""
from easygaussiansplatting_amd import gsplatcu as G


def test_key_bit_hint_is_a_decaying_maximum():
    key = ("test", 1, 2)
    G._set_key_bits(0, key, 32)
    G._learn_key_bits(0, key, 13)                  # first success after "unknown": adopt need + 1
    assert G._get_key_bits(0, key) == 14
    for _ in range(10):                            # two cameras, 13 and 14 bits, alternating: the larger one rules
        G._learn_key_bits(0, key, 14)
        assert G._get_key_bits(0, key) == 15
        G._learn_key_bits(0, key, 13)
        assert G._get_key_bits(0, key) == 15      # (need + 1 = 14 < 15: no reset, hence no miss on the next 14-bit view)
    # (the loop above ended on a 13-bit view: a run of ONE render that needed less, at most 13 + 1 bits)
    for i in range(G.KEY_BITS_DECAY - 2):          # only a long run of smaller needs lowers it ...
        G._learn_key_bits(0, key, 11 + (i % 2))
        assert G._get_key_bits(0, key) == 15
    G._learn_key_bits(0, key, 11)
    assert G._get_key_bits(0, key) == 14           # ... to the most that run needed (the 13-bit view's 13 + 1)
    G._learn_key_bits(0, key, 20, missed=True)     # a miss: full width for the redo, then adopt what it reports
    assert G._get_key_bits(0, key) == 32
    G._learn_key_bits(0, key, 20)
    assert G._get_key_bits(0, key) == 21
    G._learn_key_bits(0, key, 40)                  # (clamped)
    assert G._get_key_bits(0, key) == 32
