"""A minimal asyncio HTTP/1.1 server for ASGI apps (keep-alive, Content-Length bodies, no TLS, no websockets).
Request bodies are bounded (MAX_BODY_BYTES env, default 64 MiB: room for a 100,000-product corpus upload); a
malformed or negative Content-Length is answered 400, an oversized one 413, `Transfer-Encoding: chunked` 501 and
`Expect: 100-continue` gets its interim reply.  uvicorn stays the hardened option.  uvicorn works with this app too (`uvicorn instacart_next_order_recommendation_amd.api.app:app`,
as the reference starts its own: `uvicorn src.api.main:app`); without httptools/uvloop in the image its pure-Python
h11 parser costs more per request than FastAPI itself, so the multi-process server (serve.py) uses this one.
One asyncio.Protocol per connection, requests on a connection handled in order (pipelining is not reordered).
"""
from __future__ import annotations

import asyncio
import os
import socket
from typing import Optional

MAX_BODY_BYTES = int(os.getenv("MAX_BODY_BYTES", str(64 << 20)))

_REASON = {200: b"OK", 400: b"Bad Request", 401: b"Unauthorized", 404: b"Not Found", 405: b"Method Not Allowed",
           413: b"Payload Too Large", 422: b"Unprocessable Entity", 500: b"Internal Server Error",
           501: b"Not Implemented", 503: b"Service Unavailable"}


class _Conn(asyncio.Protocol):
    def __init__(self, app, loop):
        self.app, self.loop = app, loop
        self.buf = bytearray()
        self.transport: Optional[asyncio.Transport] = None
        self.busy = False
        self.continued = False  # "100 Continue" already sent for the request being received
        self.peer = ("0.0.0.0", 0)

    def connection_made(self, transport):
        self.transport = transport
        self.peer = transport.get_extra_info("peername") or self.peer

    def data_received(self, data: bytes):
        self.buf += data
        if not self.busy:
            self._next()

    def _next(self):
        head_end = self.buf.find(b"\r\n\r\n")
        if head_end < 0:
            if len(self.buf) > 65536:
                self.transport.close()
            return
        lines = bytes(self.buf[:head_end]).split(b"\r\n")
        try:
            method, target, _ = lines[0].split(b" ", 2)
        except ValueError:
            self.transport.close()
            return
        headers, clen, keep, expect = [], 0, True, False
        for ln in lines[1:]:
            k, _, v = ln.partition(b":")
            k, v = k.strip().lower(), v.strip()
            headers.append((k, v))
            if k == b"content-length":
                if not v.isdigit():  # empty, negative, non-numeric: the framing of everything behind it is unknown
                    return self._reject(400)
                clen = int(v)
                if clen > MAX_BODY_BYTES:
                    return self._reject(413)
            elif k == b"transfer-encoding" and v.lower() != b"identity":
                return self._reject(501)
            elif k == b"expect" and v.lower() == b"100-continue":
                expect = True
            elif k == b"connection" and v.lower() == b"close":
                keep = False
        total = head_end + 4 + clen
        if len(self.buf) < total:
            if expect and not self.continued:
                self.continued = True
                self.transport.write(b"HTTP/1.1 100 Continue\r\n\r\n")
            return
        self.continued = False
        body = bytes(self.buf[head_end + 4:total])
        del self.buf[:total]
        self.busy = True
        path, _, query = target.partition(b"?")
        scope = {"type": "http", "asgi": {"version": "3.0"}, "http_version": "1.1", "method": method.decode(),
                 "path": path.decode("latin-1"), "raw_path": path, "query_string": query, "root_path": "",
                 "scheme": "http", "headers": headers, "client": self.peer, "server": ("icrec", 0), "state": {}}
        self.loop.create_task(self._handle(scope, body, keep))

    def _reject(self, code: int) -> None:
        """Answer a request whose framing cannot be honoured and close: nothing behind it can be parsed."""
        self.buf.clear()
        if self.transport is not None:
            self.transport.write(b"HTTP/1.1 %d %s\r\ncontent-length: 0\r\nconnection: close\r\n\r\n" % (code, _REASON[code]))
            self.transport.close()

    async def _handle(self, scope, body: bytes, keep: bool):
        sent = False
        out = bytearray()
        done = asyncio.Event()

        async def receive():
            nonlocal sent
            if not sent:
                sent = True
                return {"type": "http.request", "body": body, "more_body": False}
            await done.wait()
            return {"type": "http.disconnect"}

        async def send(msg):
            if msg["type"] == "http.response.start":
                st = msg["status"]
                out.extend(b"HTTP/1.1 %d %s\r\n" % (st, _REASON.get(st, b"OK")))
                has_len = False
                for k, v in msg.get("headers") or ():
                    out.extend(k + b": " + v + b"\r\n")
                    has_len = has_len or k.lower() == b"content-length"
                self._has_len = has_len
                self._head = bytes(out)
                out.clear()
            elif msg["type"] == "http.response.body":
                out.extend(msg.get("body", b""))
                if not msg.get("more_body"):
                    head = self._head
                    if not self._has_len:
                        head += b"content-length: %d\r\n" % len(out)
                    head += b"connection: keep-alive\r\n\r\n" if keep else b"connection: close\r\n\r\n"
                    if self.transport is not None and not self.transport.is_closing():
                        self.transport.write(head + bytes(out))
                    done.set()

        try:
            await self.app(scope, receive, send)
        except Exception:  # noqa: BLE001 - the app's own handlers already turned errors into responses
            if not done.is_set() and self.transport is not None:
                self.transport.write(b"HTTP/1.1 500 Internal Server Error\r\ncontent-length: 0\r\n\r\n")
        self.busy = False
        if not keep and self.transport is not None:
            self.transport.close()
        elif self.buf:
            self._next()

    def connection_lost(self, exc):
        self.transport = None


async def serve(app, host: str = "127.0.0.1", port: int = 8000, reuse_port: bool = True, ready=None) -> None:
    """Run the ASGI lifespan, then accept connections until cancelled."""
    loop = asyncio.get_running_loop()
    sock = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
    if reuse_port:
        sock.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEPORT, 1)  # the kernel spreads connections over the workers
    sock.bind((host, port))
    sock.listen(4096)
    async with app.router.lifespan_context(app):
        server = await loop.create_server(lambda: _Conn(app, loop), sock=sock)
        if ready is not None:
            ready()
        async with server:
            await server.serve_forever()
