"""noLZSS.utils: validation and the v2 binary factor file readers (shared with nolzss_amd.utils;
reference: src/noLZSS/utils.py)."""
from nolzss_amd.utils import (NoLZSSError, InvalidInputError, validate_input, read_factors_binary_file,
                              read_binary_file_metadata, read_factors_binary_file_with_metadata)

__all__ = ["NoLZSSError", "InvalidInputError", "validate_input", "read_factors_binary_file",
           "read_binary_file_metadata", "read_factors_binary_file_with_metadata"]
