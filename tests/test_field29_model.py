"""CPU checks of the 29-bit-limb arithmetic behind k_accumulate29 (csrc/field29.cuh): the generated instruction streams
of the Montgomery product / squaring are interpreted and compared with big-integer arithmetic (incl. the 64-bit column
accumulator never overflowing at the stated operand bounds), the limb-exact model of the lazy mixed addition is run
against affine arithmetic with its value and limb bounds asserted, and the checked-in field29_asm.inc must be what the
generator emits (so the checks above are checks of the code that is compiled)."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spec = importlib.util.spec_from_file_location("gen_field29_asm", os.path.join(ROOT, "tools", "gen_field29_asm.py"))
gen = importlib.util.module_from_spec(spec)
spec.loader.exec_module(gen)


def test_generated_file_is_current():
    assert open(gen.INC_PATH).read() == gen.render()


def test_product_and_squaring_streams():
    gen.check(gen.P_FP, "Fp", trials=120)
    gen.check(gen.P_FQ, "Fq", trials=120)


def test_madd29_limb_model():
    gen.check_madd(gen.P_FP, "Fp", chains=6, length=30)
    gen.check_madd(gen.P_FQ, "Fq", chains=6, length=30)


def test_add29_limb_model():
    """the full lazy addition of the wide MSM path's bucket reduction (field29.cuh add29): random sums of sums -- operands that are themselves results of
    madd29 / add29 chains -- against affine arithmetic, value and limb bounds asserted, and the equal / opposite-point filter (A + A, the same point through
    two routes, A + (-A) must be handed to the exact path)"""
    gen.check_add(gen.P_FP, "Fp", trees=3, leaves=16)
    gen.check_add(gen.P_FQ, "Fq", trees=3, leaves=16)


def test_spread_constants_dominate_their_subtrahends():
    """a - b + K p is computed limb-wise as a_i + C_i - b_i: C must dominate every limb b can have at its stated value
    bound (madd29: x < 6p under S71, y < 4p under S51, PPP + 2Q < 3.3p with limbs <= 3 MASK under S44, rx < 5.4p under
    S61; 5p - y is formed with S51 too)."""
    for p in (gen.P_FP, gen.P_FQ):
        for (K, J), bound in zip(gen.SPREADS, (6.0, 4.0, 3.3, 5.4)):
            c = gen.spread(p, K, J)
            assert all(c[i] >= J * gen.MASK for i in range(8))
            top = int(bound * p) >> 232                      # largest top limb of a NORMALISED value below bound * p
            assert c[8] >= top + (J - 1), (K, J)             # (+ J - 1: the top limb of a J-term limb-wise sum is not normalised)
