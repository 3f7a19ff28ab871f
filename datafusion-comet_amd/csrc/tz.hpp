// Time zones for the casts and date-part functions that need one (conversion_funcs/temporal.rs, cast.rs:228-420, datetime_funcs/extract_date_part.rs):
// the reference resolves zone names with chrono-tz 0.10.4 (the IANA database compiled in); here the zone's TZif file is read from the system's
// database ($TZDIR, /usr/share/zoneinfo — what the JVM beside us resolves the same names with) and flattened into one table of (instant, offset)
// pairs that the device searches.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

namespace comet {

struct ZoneTable {
  // offset (seconds east of UTC) in force from at[i] (UTC seconds) until at[i + 1]; before at[0]: first_off
  std::vector<int64_t> at;
  std::vector<int32_t> off;
  int32_t first_off = 0;
  // the table answers instants below `limit`; later ones are read 400-year periods earlier (a zone whose rule goes on for ever is expanded over
  // the 400 years behind its last explicit transition; INT64_MAX: no rule, no end)
  int64_t limit = INT64_MAX;
  // the layout the device functions read (comet_device.hpp "time zones"): { n, first_off, limit, at[0..n), off[0..n) }
  std::vector<int64_t> flat() const;
};

// "UTC", "Z", "+05:30", "GMT-8", … : a table without transitions; region names: the zone's file.  Throws CometError (unknown zone, no database).
std::shared_ptr<const ZoneTable> load_zone(const std::string& name);
// "UTC" / "Z" / "GMT" / "Etc/UTC" / "+HH:MM" / "-HH[:MM[:SS]]" / "UTC+h" / "GMT-h" → seconds east of UTC
bool fixed_zone_offset(const std::string& tz, long long& secs);

}  // namespace comet
