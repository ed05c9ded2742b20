"""Minimal stand-in for `hydra` — TEST INFRASTRUCTURE ONLY (oracle harness)."""


def main(*args, **kwargs):
    def deco(fn):
        return fn
    return deco
