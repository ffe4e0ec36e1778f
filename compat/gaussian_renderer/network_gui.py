"""Stub of the SIBR viewer socket (interactive GUI is out of scope: SURVEY.md section 2 row 10).  PEGASUS only
touches it when GUI=True (pegasus.py:249-251,266-279); with GUI off these are never called."""
conn = None
addr = None


def init(wish_host, wish_port):
    raise NotImplementedError("the SIBR network viewer is out of scope for this build")


def try_connect():
    return None


def receive():
    raise NotImplementedError("the SIBR network viewer is out of scope for this build")


def send(message_bytes, verify):
    raise NotImplementedError("the SIBR network viewer is out of scope for this build")
