"""``fewbit.compat``: the one helper the reference keeps for old interpreters (fewbit/compat.py; its linear tests import
``removeprefix`` from here).  Python >= 3.9 has it as a str method."""

__all__ = ['removeprefix']


def removeprefix(text: str, prefix: str, /) -> str:
    """``text`` without a leading ``prefix`` (``str.removeprefix``)."""
    return text.removeprefix(prefix)
