"""Constructor-signature introspection used by ``get_pooler`` (reference: tgp/utils/signature.py)."""
import inspect
from typing import Callable, List, NamedTuple, Type, Union


class Signature(NamedTuple):
    args: List[str]
    has_varargs: bool
    has_kwargs: bool


def foo_signature(foo: Union[Callable, Type]) -> Signature:
    target = foo.__init__ if isinstance(foo, type) else foo
    spec = inspect.getfullargspec(target)
    names = list(spec.args)
    if names and names[0] in ("self", "cls"):
        names = names[1:]
    return Signature(args=names, has_varargs=spec.varargs is not None, has_kwargs=spec.varkw is not None)
