"""Variant specs for the A/B tools: `path/to/lib.so` or `path/to/lib.so@ENV=VAL,ENV2=VAL` (environment tunables read by the
library's launchers through od_env_int); an empty path before `@` means the in-tree library."""
import os

DEFAULT_LIB = os.path.join("osu_dreamer_amd", "libosudreamer_hip.so")


def parse(spec):
    lib, _, envs = spec.partition("@")
    lib = lib or DEFAULT_LIB
    env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
    label = os.path.basename(lib) + ("@" + envs if envs else "")
    return label, os.path.abspath(lib), env
