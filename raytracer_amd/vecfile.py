"""Reader / writer of the datapath vector-file format (.vec).

Text, one stream per file:
    # comment
    S <suite name> extended=<0|1>
    I <84 hex words>      one rt_dp_input_bundle  (EnhancedInputBundle image)
    O <21 hex words>      the rt_dp_output_bundle expected for the record above
Words are 32-bit, hexadecimal, without leading zeros, in struct order
(include/rt_dp.h).  A fresh datapath (accumulators +0) is assumed at the start.
The same format is consumed by the rt_dp_replay host tool.
"""
import numpy as np

IN_WORDS, OUT_WORDS = 84, 21


def read(path):
    """Returns (name, extended, inputs [n, 84] uint32, expected [n, 21] uint32)."""
    name, extended, ins, outs = "", 1, [], []
    with open(path) as f:
        for line in f:
            if not line.strip() or line[0] == "#":
                continue
            tag, rest = line[0], line[1:].split()
            if tag == "S":
                name = rest[0]
                for tok in rest[1:]:
                    if tok.startswith("extended="):
                        extended = int(tok.split("=")[1])
            elif tag == "I":
                assert len(rest) == IN_WORDS, (path, len(rest))
                ins.append([int(x, 16) for x in rest])
            elif tag == "O":
                assert len(rest) == OUT_WORDS, (path, len(rest))
                outs.append([int(x, 16) for x in rest])
            else:
                raise ValueError("%s: unknown record tag %r" % (path, tag))
    assert len(ins) == len(outs), path
    return name, extended, np.array(ins, np.uint32).reshape(-1, IN_WORDS), np.array(outs, np.uint32).reshape(-1, OUT_WORDS)


def write(path, name, extended, inputs, outputs, comment=None):
    with open(path, "w") as f:
        if comment:
            f.write("# %s\n" % comment)
        f.write("S %s extended=%d\n" % (name, extended))
        for i in range(inputs.shape[0]):
            f.write("I " + " ".join("%x" % x for x in inputs[i]) + "\n")
            f.write("O " + " ".join("%x" % x for x in outputs[i]) + "\n")
