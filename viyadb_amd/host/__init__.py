"""C++ host shim (mirror of the reference's db/query interfaces for the aggregate path)."""
