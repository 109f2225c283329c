"""CPU: the lane table of the one-state-per-wave Keccak (falcon_amd/csrc/keccak_lanes.hpp, used by hash_wave_body in
tiny_kernel.hpp) against the textbook permutation: tests/cpp/keccak_lanes_check emulates the GPU's round lane by lane
(row-wise shifted XORs, lane fetches, funnel shifts) on random states and compares all 24 rounds with Keccak-f[1600]
on 64-bit words (FIPS 202; the permutation under hashing::hash_to_point, hashing.hpp:30-39)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def test_keccak_lane_table_matches_textbook_permutation():
    subprocess.run(["make", "-C", CPP, "keccak_lanes_check"], check=True, capture_output=True, timeout=120)
    out = subprocess.run([os.path.join(CPP, "keccak_lanes_check")], capture_output=True, text=True, timeout=60)
    assert out.returncode == 0 and out.stdout.strip() == "ok", out.stdout + out.stderr
