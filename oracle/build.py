"""Builds the checker: the plain-C restatement (oracle/libkmer_oracle.so) and, where the reference sources are present
(the build container only), the compiled reference under oracle/_ref/.  Test infrastructure: the product package
(kmer_id_amd/) neither builds nor loads any of this."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build_oracle(verbose=False):
    subprocess.check_call(["make", "-C", HERE, "oracle"] + ([] if verbose else ["-s"]))
    if os.path.exists("/root/reference/newkmer_10nx.cpp"):
        subprocess.check_call(["make", "-C", HERE, "ref"] + ([] if verbose else ["-s"]))
    return os.path.join(HERE, "libkmer_oracle.so")


if __name__ == "__main__":
    build_oracle(verbose=True)
