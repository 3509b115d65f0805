#!/usr/bin/env python3
"""Prints the radix-2^29 constants embedded in crypto_primitives_amd/csrc/f29.hpp (host big-int; run by hand)."""
P = 52435875175126190479447740508185965837690552500527637822603658699938581184513
D = (-10240 * pow(10241, -1, P)) % P
R = 1 << 261


def limbs(x, n=9):
    assert 0 <= x < (1 << (29 * n))
    return ", ".join("0x%08xu" % ((x >> (29 * i)) & ((1 << 29) - 1)) for i in range(n))


print("P29      ", limbs(P))
print("ONE (R mod p)            ", limbs(R % P))
print("K_IN  = 2^266 mod p      ", limbs(pow(2, 266, P)))
print("K_OUT = 2^256 mod p      ", limbs(pow(2, 256, P)))
print("R2    = 2^522 mod p      ", limbs(pow(2, 522, P)))
print("TE_D  = d * R mod p      ", limbs(D * R % P))
print("4P                       ", limbs(4 * P))
print("2P                       ", limbs(2 * P))
print("-p^-1 mod 2^29 =", (-pow(P, -1, 1 << 29)) % (1 << 29), "(== 2^29-1 means m = -t0)")
