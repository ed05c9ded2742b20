"""typeguard stand-in (TEST INFRASTRUCTURE ONLY): check_type is a no-op."""


def check_type(*args, **kwargs):
    return True
