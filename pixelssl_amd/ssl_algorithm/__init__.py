"""Registry of the SSL algorithms (pixelssl/ssl_algorithm/__init__.py:10-27): module name == NAME ==
export-function name; looked up as `ssl_algorithm.__dict__[name].__dict__[name]`."""
from . import ssl_base, ssl_null, ssl_mt, ssl_adv, ssl_cutmix, ssl_gct, ssl_cct, ssl_s4l

SSL_NULL = ssl_null.SSLNULL.NAME
SSL_MT = ssl_mt.SSLMT.NAME
SSL_ADV = ssl_adv.SSLADV.NAME
SSL_CUTMIX = ssl_cutmix.SSLCUTMIX.NAME
SSL_GCT = ssl_gct.SSLGCT.NAME
SSL_CCT = ssl_cct.SSLCCT.NAME
SSL_S4L = ssl_s4l.SSLS4L.NAME

SSL_ALGORITHMS = [SSL_NULL, SSL_MT, SSL_ADV, SSL_CUTMIX, SSL_GCT, SSL_CCT, SSL_S4L]
