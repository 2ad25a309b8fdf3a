"""Exception types of the host adapter; names and meaning mirror gym/error.py of the reference."""


class Error(Exception):
    """Base error (gym/error.py:5); raised e.g. for a negative seed (gym/utils/seeding.py:21-22)."""


class UnregisteredEnv(Error):
    """Unknown env id (gym/error.py:15)."""


class ResetNeeded(Error):
    """step() before reset() (gym/error.py:56, gym/wrappers/order_enforcing.py:33-37)."""


class InvalidAction(Error):
    """Action not contained in the action space (gym/error.py:77)."""


class AlreadyPendingCallError(Error):
    """step_async() while a step is pending (gym/error.py:171)."""

    def __init__(self, message: str, name: str = ""):
        Exception.__init__(self, message)   # not super(): interop may put gym.error's class (message, name) next in the MRO
        self.name = name


class NoAsyncCallError(Error):
    """step_wait() without step_async() (gym/error.py:180)."""

    def __init__(self, message: str, name: str = ""):
        Exception.__init__(self, message)   # not super(): interop may put gym.error's class (message, name) next in the MRO
        self.name = name


class ClosedEnvironmentError(Error):
    """Operation on a closed vector env (gym/error.py:189)."""


class CustomSpaceError(Error):
    """A space that is none of the built-in kinds reached a batching helper (gym/error.py:190-197, gym/vector/utils/spaces.py:205-212)."""
