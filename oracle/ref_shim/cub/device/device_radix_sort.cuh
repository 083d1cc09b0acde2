// stub: hipcub.hpp (pulled in by the hipified include above it) already provides DeviceRadixSort
