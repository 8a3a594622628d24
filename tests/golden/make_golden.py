#!/usr/bin/env python3
"""Extracts the reference's own golden vectors for the MSM/NTT hot path into
small JSON fixtures that can travel to the GPU box (which has no
/root/reference).  Run in the build container:

    python tests/golden/make_golden.py [/root/reference]

Sources (all under /root/reference):
  kimchi/src/proof.rs:1160-1204            MSM known-answer (Vesta, 16 terms)
  poly-commitment/tests/commitment.rs:289-345  trusted-setup SRS bytes (both curves)
  poly-commitment/tests/commitment.rs:348-386  chunked+masked commit bytes
  srs/vesta.srs, srs/pallas.srs            65,536 g_i + h of SRS::create(1<<16)
  curves/src/pasta/fields/{fp,fq}.rs       MODULUS / R / R2 / INV / 2-adic root

Nothing here is computed by this repository's code: values are copied
(parsed) from the reference's sources and binary fixtures, so the oracle and
kernels can be pinned against them.
"""
import hashlib
import json
import os
import re
import struct
import sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def read(rel):
    with open(os.path.join(REF, rel), "r") as f:
        return f.read()


def vec_u8_after(src: str, fn_name: str, nth: int = 0):
    """The nth `let buf_expected: Vec<u8> = vec![ ... ];` inside fn `fn_name`."""
    start = src.index("fn " + fn_name)
    end = src.find("\n}\n", start)
    body = src[start:end]
    blocks = re.findall(r"buf_expected: Vec<u8> = vec!\[(.*?)\];", body, re.S)
    return [int(x) for x in re.findall(r"\d+", blocks[nth])]


def bigint_limbs(src: str, const_name: str):
    m = re.search(r"const %s: BigInteger =?\s*BigInteger::new\(\[(.*?)\]\)" % const_name, src, re.S)
    limbs = [int(x, 16) for x in re.findall(r"0x[0-9a-fA-F]+", m.group(1))]
    assert len(limbs) == 4
    return sum(l << (64 * i) for i, l in enumerate(limbs))


def field_consts(rel):
    src = read(rel)
    out = {}
    out["modulus_dec"] = re.search(r'#\[modulus = "(\d+)"\]', src).group(1)
    for k in ("TWO_ADIC_ROOT_OF_UNITY", "MODULUS", "R", "R2", "T", "GENERATOR"):
        out[k] = hex(bigint_limbs(src, k))
    out["INV"] = re.search(r"const INV: u64 = (\d+);", src).group(1)
    return out


def srs_file(rel):
    with open(os.path.join(REF, rel), "rb") as f:
        data = f.read()
    assert data[0] == 0x92 and data[1] == 0xDD
    n = struct.unpack(">I", data[2:6])[0]
    pts = [data[6 + 35 * i + 2: 6 + 35 * i + 35] for i in range(n)]
    off = 6 + 35 * n
    h = data[off + 2: off + 35]
    assert off + 35 == len(data)
    sample_idx = list(range(64)) + [1000, 4095, 4096, 32767, 32768, 65534, 65535]
    prefix_digest = {}
    for k in range(0, 17):
        m = 1 << k
        prefix_digest[str(k)] = hashlib.blake2b(b"".join(pts[:m]), digest_size=32).hexdigest()
    return {
        "n": n,
        "h": h.hex(),
        "samples": {str(i): pts[i].hex() for i in sample_idx},
        # blake2b-256 over the concatenated 33-byte compressed points g_0..g_{2^k-1}
        "prefix_digest_blake2b256": prefix_digest,
    }


def main():
    proof_rs = read("kimchi/src/proof.rs")
    i0 = proof_rs.index("fn test_recursion_challenge_commitment_regression")
    seg = proof_rs[i0:i0 + 3000]
    xs = re.findall(r'Fq::from_str\(\s*"(\d+)"', seg)
    chals = [int(x) for x in re.findall(r"Fp::from\((\d+)u64\)", seg)][:4]
    vesta_rs = read("curves/src/pasta/curves/vesta.rs")
    pallas_rs = read("curves/src/pasta/curves/pallas.rs")
    gy_v = re.search(r'G_GENERATOR_Y: Fq =\s*MontFp!\("(\d+)"\)', vesta_rs).group(1)
    gy_p = re.search(r'G_GENERATOR_Y: Fp =\s*MontFp!\("(\d+)"\)', pallas_rs).group(1)

    ctest = read("poly-commitment/tests/commitment.rs")
    golden = {
        "_generated_by": "tests/golden/make_golden.py from the reference snapshot at /root/reference",
        "fields": {
            "Fp": field_consts("curves/src/pasta/fields/fp.rs"),
            "Fq": field_consts("curves/src/pasta/fields/fq.rs"),
        },
        "generators": {"vesta": ["1", gy_v], "pallas": ["1", gy_p]},
        "msm_kat": {  # kimchi/src/proof.rs:1160-1204
            "curve": "vesta",
            "chals": chals,
            "basis": "(i+1)*G for i in 0..16",
            "expected_x": xs[0],
            "expected_y": xs[1],
        },
        "srs_trusted_setup_kat": {  # tests/commitment.rs:289-345, seed [0;32], depth 8
            "vesta": vec_u8_after(ctest, "ser_regression_canonical_srs", 0),
            "pallas": vec_u8_after(ctest, "ser_regression_canonical_srs", 1),
        },
        "commit_kat": {  # tests/commitment.rs:348-386
            "curve": "vesta", "srs_depth": 128, "com_length": 300, "num_chunks": 6,
            "seed": [0] * 32,
            "bytes": vec_u8_after(ctest, "ser_regression_canonical_polycomm", 0),
        },
        "opening_proof_kat": {  # tests/commitment.rs:388-440: proofs[0] of generate_random_opening_proof, seed [0;32], SRS 2^7
            "curve": "vesta", "srs_depth": 128, "seed": [0] * 32,
            "bytes": vec_u8_after(ctest, "ser_regression_canonical_opening_proof", 0),
        },
        "srs": {
            "vesta": srs_file("srs/vesta.srs"),
            "pallas": srs_file("srs/pallas.srs"),
        },
    }
    with open(os.path.join(OUT, "reference_kats.json"), "w") as f:
        json.dump(golden, f, indent=1)
    print("wrote", os.path.join(OUT, "reference_kats.json"))
    whole_proof_kat()


def whole_proof_kat():
    """kimchi/src/tests/and.rs:126-160, 404-731: the serialised ProverProof of the 8-byte AND gadget proved with
    make_test_rng(Some(RNG_SEED)) -- the reference's only byte-exact whole-proof regression.  Seed and bytes are parsed from
    the test's source; written as and_serialization_regression.json (seed, bytes as hex)."""
    src = read("kimchi/src/tests/and.rs")
    body = src[src.index("fn prove_and_check_serialization_regression"):]
    seed = [int(x) for x in re.findall(r"\d+", re.search(r"const RNG_SEED: \[u8; 32\] = \[(.*?)\];", body, re.S).group(1))]
    reg = src[src.index("fn test_serialization_regression"):]
    buf = bytes(int(x) for x in re.findall(r"\d+", re.search(r"let buf_expected = vec!\[(.*?)\];", reg, re.S).group(1)))
    assert len(seed) == 32 and len(buf) == 6160
    with open(os.path.join(OUT, "and_serialization_regression.json"), "w") as f:
        json.dump({"source": "kimchi/src/tests/and.rs:126-160,404-731", "curve": "vesta", "bytes_of_and": 8, "seed": seed,
                   "sha256": hashlib.sha256(buf).hexdigest(), "proof_hex": buf.hex()}, f, indent=1)
    print("wrote and_serialization_regression.json")


if __name__ == "__main__":
    main()
