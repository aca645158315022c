"""Alias registry: the JSON/YAML configuration surface of the hot path.

Mirrors the behaviour of the reference's ``pydrobert.speech.alias`` (alias.py:28-100):
every pluggable type (frame computers, filter banks, windows, scales, post-processors)
derives from :class:`AliasedFactory` and is looked up by a string alias; a mapping with
the key ``"alias"`` (else ``"name"``) plus constructor keyword arguments is the config
format.  Resolution order is the reference's (alias.py:58-69): descendants are searched
before the class itself and the most recently defined subclass first, so a class defined
later shadows an earlier one with the same alias -- the hook by which the HIP-backed
``stft`` computer can replace a numpy one.
"""
import abc
from typing import Any, Mapping, Optional, Set, Type, TypeVar, Union

__all__ = ["alias_factory_subclass_from_arg", "AliasedFactory"]

T = TypeVar("T", bound="AliasedFactory")


def _resolve(klass: type, alias: str) -> Optional[type]:
    # post-order walk, newest child first (reference: alias.py:58-68 pops the stack's
    # tail, i.e. the last element of __subclasses__(), and only tests a class after
    # all of its descendants)
    for child in reversed(klass.__subclasses__()):
        found = _resolve(child, alias)
        if found is not None:
            return found
    if alias in klass.aliases:
        return klass
    return None


class AliasedFactory(abc.ABC):
    """Base of every type that can be built from an alias"""

    aliases: Set[str] = set()

    @classmethod
    def from_alias(cls: Type[T], alias: str, *args, **kwargs) -> T:
        """Instantiate the subclass (or this class) registered under `alias`

        Raises
        ------
        ValueError
            If no class in the hierarchy rooted at `cls` carries `alias`
        """
        target = _resolve(cls, alias)
        if target is None:
            raise ValueError(f"Cannot find subclass with alias '{alias}'")
        return target(*args, **kwargs)


def alias_factory_subclass_from_arg(
    factory_class: Type[T], arg: Union[T, str, Mapping[str, Any]]
) -> T:
    """Turn an instance / alias string / config mapping into an instance

    Same three cases as the reference (alias.py:90-100): an instance of
    `factory_class` is returned untouched; a string is an alias with no arguments;
    anything else is copied to a dict whose ``"alias"`` (or, failing that,
    ``"name"``) entry names the class and whose other entries are keyword arguments.
    """
    if isinstance(arg, factory_class):
        return arg
    if isinstance(arg, str):
        return factory_class.from_alias(arg)
    kwargs = dict(arg)
    if "alias" in kwargs:
        alias = kwargs.pop("alias")
    else:
        alias = kwargs.pop("name")  # KeyError if neither, as in the reference
    return factory_class.from_alias(alias, **kwargs)
