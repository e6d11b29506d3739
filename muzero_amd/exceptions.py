"""Exception family of the reference (exceptions.py:1-30) plus the error raised when the HIP library fails."""


class MuZeroError(Exception):
    """Base class for exceptions (reference exceptions.py:1-10)."""

    def __init__(self, message: str) -> None:
        super().__init__(message)
        self.message = message

    def __repr__(self) -> str:
        return self.message


class MuZeroImplementationError(MuZeroError):
    """Un-implemented methods in base classes (reference exceptions.py:13-21)."""

    def __init__(self, function_name: str, class_name: str):
        super().__init__(message=f'function {function_name} needs to be implemented by {class_name}')
        self.function_name = function_name
        self.class_name = class_name


class MuZeroEnvironmentError(MuZeroError):
    """Environment failures (reference exceptions.py:24-30)."""

    def __init__(self, message: str):
        super().__init__(message=message)


class MuZeroHipError(MuZeroError):
    """A libmzhip.so call returned a negative mz_status; carries the status and mz_last_error()."""

    def __init__(self, status: int, message: str):
        super().__init__(message=f'libmzhip status {status}: {message}')
        self.status = status
