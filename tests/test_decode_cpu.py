"""CPU: the test decoder (tests/zdw_decode.py) read against the reference's own outputs -- every vector the
compiled reference encoded decodes to what its source says; this is the independent yardstick the GPU
round-trip tests use."""
from oracle import oracle
from tests import parity_util as pu
from tests import zdw_decode
from zdw_amd import synth


def test_reference_outputs_decode_to_their_sources():
    n = 0
    for fname in ("edge_cases.json", "fuzz_cases.json"):
        for c in pu.cases(fname):
            if c["exit_code"] != 0 or not c["zdw"]:
                continue
            z, rows = zdw_decode.decode(c["zdw"])
            src = zdw_decode.interpret(list(z.types), c["tsv"], trim="-t" in c["args"])
            assert len(rows) == len(src), c["name"]
            assert all(zdw_decode.same(d, s) for d, s in zip(rows, src)), c["name"]
            n += 1
    assert n >= 200


def test_multi_block_oracle_output_decodes():
    desc, types, tsv = synth.make_table(seed=41, ncols=14, nrows=2500, vocab=3000)
    rc, out, info = oracle.encode(desc, tsv, split_rows=[700, 900], split_cols=[3, 0])
    assert rc == 0 and info.nblocks == 3
    z, rows = zdw_decode.decode(out)
    src = zdw_decode.interpret(types, tsv)
    assert len(rows) == len(src) == 2500 and all(zdw_decode.same(d, s) for d, s in zip(rows, src))
