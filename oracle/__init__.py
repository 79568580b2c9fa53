"""Test infrastructure: CPU oracle for the VTP hot path.  Never imported by vtp_amd/."""
