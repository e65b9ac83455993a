"""engine subpackage."""
