from ..layers import to_2tuple, trunc_normal_, DropPath  # noqa: F401
